"""Deployment-side pieces of the hot path: the fused detection post-processor and the wrapper
that bakes it behind the model (reference `src/dl/export.py:20-128`).  The reference's ONNX ->
TensorRT / OpenVINO exporters are NVIDIA/Intel tool-chains and out of scope (SURVEY.md section 2
row 13); `export_torchscript` writes a TorchScript-free `torch.export`-style artefact instead: the
plain state dict plus the post-processor configuration.
"""
from pathlib import Path

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels
from ..d_fine.dfine import build_model


class DFINEPostProcessor(nn.Module):
    """sigmoid -> top-K over Q*C -> (label = idx % C, query = idx // C) -> absolute xyxy boxes.
    Returns (labels [B,K] i64, boxes [B,K,4] f32, scores [B,K] f32[, masks [B,K,Hm,Wm]])."""

    def __init__(self, num_classes: int, num_top_queries: int = 300, use_focal_loss: bool = True):
        super().__init__()
        self.num_classes, self.num_top_queries = num_classes, num_top_queries
        self.use_focal_loss = use_focal_loss

    @staticmethod
    def norm_xywh_to_abs_xyxy(boxes: torch.Tensor, height: int, width: int, to_round=True) -> torch.Tensor:
        cx, cy = boxes[:, 0] * width, boxes[:, 1] * height
        hw, hh = boxes[:, 2] * width / 2, boxes[:, 3] * height / 2
        x0, y0, x1, y1 = cx - hw, cy - hh, cx + hw, cy + hh
        if to_round:      # floor/ceil and keep one pixel of margin, like the reference (export.py:49-53)
            x0, y0 = torch.clamp(torch.floor(x0), min=1), torch.clamp(torch.floor(y0), min=1)
            x1 = torch.clamp(torch.ceil(x1), max=width - 1)
            y1 = torch.clamp(torch.ceil(y1), max=height - 1)
        else:
            x0, y0 = torch.clamp(x0, min=0), torch.clamp(y0, min=0)
            x1, y1 = torch.clamp(x1, max=width), torch.clamp(y1, max=height)
        return torch.stack([x0, y0, x1, y1], dim=1)

    def forward(self, outputs: dict, input_h: int, input_w: int):
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        masks = outputs.get("pred_masks", None)
        if self.use_focal_loss:
            k = min(self.num_top_queries, logits.shape[1] * logits.shape[2])
            labels, qidx, out_boxes, scores = kernels.detection_topk(logits, boxes, k, input_h, input_w)   # one HIP kernel
        else:
            b, q = boxes.shape[:2]
            abs_boxes = self.norm_xywh_to_abs_xyxy(boxes.flatten(0, 1), input_h, input_w).view(b, q, 4)
            probs = F.softmax(logits, dim=-1)[:, :, :-1]
            scores, labels = probs.max(dim=-1)
            k = min(self.num_top_queries, scores.shape[1])
            scores, qidx = torch.topk(scores, k, dim=-1)
            labels = labels.gather(1, qidx)
            out_boxes = abs_boxes.gather(1, qidx.unsqueeze(-1).expand(-1, -1, 4))
        out = (labels, out_boxes, scores)
        if masks is not None:
            hm, wm = masks.shape[2:]
            out = out + (masks.gather(1, qidx[..., None, None].expand(-1, -1, hm, wm)),)
        return out


class ExportWrapper(nn.Module):
    def __init__(self, model: nn.Module, postprocessor: DFINEPostProcessor, input_size):
        super().__init__()
        self.model, self.postprocessor = model, postprocessor
        self.input_h, self.input_w = input_size[0], input_size[1]

    def forward(self, x):
        return self.postprocessor(self.model(x), self.input_h, self.input_w)


def prepare_model(cfg, device):
    """cfg: mapping with model_name, task, train.{label_to_name,img_size,path_to_save}."""
    model = build_model(cfg["model_name"], len(cfg["train"]["label_to_name"]),
                        enable_mask_head=cfg.get("task", "detect") == "segment", device=device,
                        img_size=cfg["train"]["img_size"])
    model.load_state_dict(torch.load(Path(cfg["train"]["path_to_save"]) / "model.pt", weights_only=True))
    return model.eval()


def export_artifact(model, num_classes, img_size, path):
    """Self-describing checkpoint for the HIP runtime: weights + post-processor configuration."""
    torch.save({"state_dict": model.state_dict(), "num_classes": num_classes, "img_size": list(img_size),
                "postprocessor": {"num_top_queries": 300, "use_focal_loss": True}}, path)
    return path


def main(argv=None):
    """`python -m custom_d_fine_amd.dl.export [key=value ...]` (alias `python -m src.dl.export`): the reference's export entry
    point (src/dl/export.py:278-334) for this stack - loads `<train.path_to_save>/model.pt`, deploys the model and writes
    `<train.path_to_save>/model.pt2`, a `torch.export` ExportedProgram of model + post-processor whose launches go through
    libdfine_hip.so (dl/export_program.py), next to the self-describing weight artefact `model_hip.pt`.  Keys: the train
    config's (`model_name`, `task`, `train.img_size`, `train.num_classes` or `train.label_to_name`, `train.path_to_save`) and
    `export.half` (bf16 autocast, default true), `export.max_batch_size` (the exported batch size, default 1)."""
    import sys
    from .train import load_config
    from .export_program import export_program
    cfg = load_config(sys.argv[1:] if argv is None else argv)
    t = cfg["train"]
    ex = cfg.get("export", {}) or {}
    if not torch.cuda.is_available():
        raise SystemExit("export traces the HIP-backed forward and needs the MI355X")
    if "label_to_name" not in t:
        t["label_to_name"] = {i: str(i) for i in range(t["num_classes"])}
    model = prepare_model(cfg, "cuda")
    n_cls = len(t["label_to_name"])
    out = Path(t["path_to_save"])
    export_artifact(model, n_cls, t["img_size"], out / "model_hip.pt")
    path, ep = export_program(model, n_cls, t["img_size"], out / "model.pt2", batch=int(ex.get("max_batch_size", 1)),
                              half=bool(ex.get("half", True)))
    print(f"exported {path} ({path.stat().st_size >> 20} MiB)")
    return path


if __name__ == "__main__":
    main()
