"""`Torch_model`: the inference runtime of the reference (`src/infer/torch_model.py:13-375`) on the HIP forward path.

Same constructor arguments, same call contract (`model(img_bgr_hwc_uint8 | batch_bhwc) -> [{"labels", "boxes", "scores"
[, "masks"]}]`, boxes absolute xyxy in the ORIGINAL image frame), same checkpoint loading (`build_model(..., img_size=None)`,
`load_state_dict(strict=False)`).  What differs is where the work runs:
  * the raw uint8 frame is uploaded as it is; resize / letterbox (OpenCV's 8-bit INTER_LINEAR arithmetic), BGR->RGB,
    HWC->CHW and /255 are ONE HIP kernel (csrc/postproc.hip: preprocess_kernel) instead of cv2 + numpy on the host;
  * sigmoid / top-K / label split is the HIP post-processor kernel (postprocess_kernel), the box mapping a few broadcast
    tensor ops; per-image Python work is limited to slicing the kept detections.
`half=True` runs the network under bf16 autocast (the MFMA path; the reference's fp16 has no counterpart on this stack).
"""
from typing import Dict, List, Tuple

import numpy as np
import torch

from .. import kernels
from ..d_fine.dfine import build_model


def letterbox_geometry(shape, new_shape, stride=32, auto=False, scaleup=True):
    """(resized (h, w), (top, left)) of the reference's letterbox(im, new_shape, auto=...) (torch_model.py:378-418)."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))          # (w, h)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    return (new_unpad[1], new_unpad[0]), (int(np.floor(dh)), int(np.floor(dw))), (int(np.floor(dh)) + int(np.ceil(dh)),
                                                                                 int(np.floor(dw)) + int(np.ceil(dw)))


def non_max_suppression(boxes, scores, labels, masks=None, iou_threshold=0.5):
    """Class-aware greedy NMS (the reference offsets boxes by class and calls torchvision.ops.nms)."""
    if boxes.numel() == 0:
        return boxes, scores, labels, masks
    off = labels.to(boxes.dtype)[:, None] * (boxes.max() + 1)
    b = boxes + off
    order = scores.argsort(descending=True)
    b = b[order]
    area = (b[:, 2] - b[:, 0]).clamp(min=0) * (b[:, 3] - b[:, 1]).clamp(min=0)
    lt = torch.maximum(b[:, None, :2], b[None, :, :2])
    rb = torch.minimum(b[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
    iou_c = iou.cpu()
    keep, dead = [], torch.zeros(len(b), dtype=torch.bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        dead |= iou_c[i] > iou_threshold
    idx = order[torch.tensor(keep, device=order.device)]
    return boxes[idx], scores[idx], labels[idx], (masks[idx] if masks is not None else None)


class Torch_model:
    def __init__(self, model_name: str, model_path: str, n_outputs: int, input_width: int = 640, input_height: int = 640,
                 conf_thresh: float = 0.5, rect: bool = False, half: bool = False, keep_ratio: bool = False,
                 use_nms: bool = False, enable_mask_head: bool = False, binarize_masks: bool = True,
                 mask_threshold: float = 0.5, device: str = None, hip_graph: bool = False):
        """hip_graph: capture the network forward of every (batch, input size) it meets into a HIP graph and replay it -
        one launch instead of ~1 000 host-issued kernel launches per call (the eager forward is host-bound: ~11 ms per call
        from batch 1 to 16, profiles/r03_infer_table.txt).  The weights must not change afterwards (inference)."""
        self.input_size = (input_height, input_width)
        self.n_outputs, self.model_name, self.model_path = n_outputs, model_name, model_path
        self.rect, self.half, self.keep_ratio, self.use_nms = rect, half, keep_ratio, use_nms
        self.enable_mask_head, self.binarize_masks, self.mask_threshold = enable_mask_head, binarize_masks, mask_threshold
        self.channels = 3
        self.conf_threshs = [conf_thresh] * n_outputs if isinstance(conf_thresh, float) else list(conf_thresh)
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self._thr = None
        self.hip_graph = bool(hip_graph) and str(self.device).startswith("cuda")
        self.max_graphs = 8                              # captured input shapes kept (least recently used dropped)
        self._graphs = {}
        self._load_model()
        self._test_pred()

    # ------------------------------------------------------------------------------------------------ model
    def _load_model(self):
        self.model = build_model(self.model_name, self.n_outputs, self.enable_mask_head, self.device, img_size=None)
        if self.model_path is not None:
            self.model.load_state_dict(torch.load(self.model_path, weights_only=True, map_location="cpu"), strict=False)
        self.model.eval().to(self.device)
        if str(self.device).startswith("cuda"):
            # the weights are fixed from here on: eval-mode BatchNorm folds are formed once, so that a conv -> BN -> act unit of the
            # bf16 forward is a single launch (kernels.conv_bn_act: affine + activation in the convolution's store phase)
            from .. import kernels
            kernels.freeze_eval_affine(self.model)

    def _test_pred(self) -> None:
        img = np.random.randint(0, 255, size=(1100, 1000, self.channels), dtype=np.uint8)
        self(img)

    # ------------------------------------------------------------------------------------------------ input
    def _compute_nearest_size(self, shape, target_size, stride=32) -> Tuple[int, int]:
        scale = target_size / max(shape)
        new_shape = [int(round(dim * scale)) for dim in shape]
        return [max(stride, int(np.ceil(dim / stride) * stride)) for dim in new_shape]

    def _geometry(self, h, w):
        """output (H, W), resized (h, w), (top, left) for a source frame of h x w."""
        if not self.keep_ratio:
            return self.input_size, self.input_size, (0, 0)
        if self.rect:
            out = tuple(self._compute_nearest_size((h, w), max(*self.input_size)))
        else:
            out = self.input_size
        resized, tl, _ = letterbox_geometry((h, w), out, auto=False)
        return out, resized, tl

    def _prepare_inputs(self, inputs):
        """uint8 BGR [H, W, 3] or [B, H, W, 3] (numpy or torch) -> (float [B, 3, Hn, Wn] on the device, processed sizes,
        original sizes).  One H2D copy of the raw frame(s) and one kernel."""
        frames = torch.as_tensor(inputs) if not torch.is_tensor(inputs) else inputs
        if frames.dim() == 3:
            frames = frames[None]
        assert frames.dtype == torch.uint8 and frames.shape[-1] == 3
        B, h, w, _ = frames.shape
        frames = frames.to(self.device, non_blocking=True).contiguous()
        out_hw, resized, tl = self._geometry(h, w)
        x = kernels.preprocess_frames(frames, out_hw, resized, tl)
        return x, [tuple(out_hw)] * B, [(h, w)] * B

    # ------------------------------------------------------------------------------------------------ output
    @staticmethod
    def process_boxes(boxes, processed_sizes, orig_sizes, keep_ratio):
        """[B, Q, 4] normalised cxcywh -> absolute xyxy in the original frames, fp32 throughout like the reference's torch
        version (torch_model.py:87-102, 455-480)."""
        dev = boxes.device
        ps = torch.tensor(processed_sizes, device=dev, dtype=torch.float32)          # [B, 2] (h, w)
        osz = torch.tensor(orig_sizes, device=dev, dtype=torch.float32)
        ph, pw = ps[:, 0:1], ps[:, 1:2]
        b = boxes.float()
        xc, yc, bw, bh = b[..., 0] * pw, b[..., 1] * ph, b[..., 2] * pw, b[..., 3] * ph
        x0 = torch.clamp(torch.floor(xc - bw / 2), min=1)
        y0 = torch.clamp(torch.floor(yc - bh / 2), min=1)
        x1 = torch.minimum(torch.ceil(xc + bw / 2), pw - 1)
        y1 = torch.minimum(torch.ceil(yc + bh / 2), ph - 1)
        oh, ow = osz[:, 0:1], osz[:, 1:2]
        if keep_ratio:
            g64 = torch.minimum(ps[:, 0].double() / osz[:, 0].double(), ps[:, 1].double() / osz[:, 1].double())
            padw = torch.round((ps[:, 1].double() - osz[:, 1].double() * g64) / 2 - 0.1).float()[:, None]
            padh = torch.round((ps[:, 0].double() - osz[:, 0].double() * g64) / 2 - 0.1).float()[:, None]
            gain = g64.float()[:, None]
            x0, x1 = (x0 - padw) / gain, (x1 - padw) / gain
            y0, y1 = (y0 - padh) / gain, (y1 - padh) / gain
            zero = torch.zeros_like(ow)
            x0, x1 = torch.maximum(torch.minimum(x0, ow), zero), torch.maximum(torch.minimum(x1, ow), zero)
            y0, y1 = torch.maximum(torch.minimum(y0, oh), zero), torch.maximum(torch.minimum(y1, oh), zero)
        else:
            sx, sy = ow / pw, oh / ph
            x0, x1, y0, y1 = x0 * sx, x1 * sx, y0 * sy, y1 * sy
        return torch.stack([x0, y0, x1, y1], dim=-1)

    @staticmethod
    def process_masks(pred_masks, processed_size, orig_sizes, keep_ratio) -> List[torch.Tensor]:
        """[B, Q, Hm, Wm] -> list of [Q, H0, W0] in [0, 1] (letterbox pad removed when keep_ratio; torch_model.py:104-157)."""
        single = pred_masks.dim() == 3
        if single:
            pred_masks = pred_masks.unsqueeze(0)
        _, _, hm, wm = pred_masks.shape
        proc_h, proc_w = int(processed_size[0]), int(processed_size[1])
        out = []
        for b in range(pred_masks.shape[0]):
            h0, w0 = int(orig_sizes[b][0]), int(orig_sizes[b][1])
            m = pred_masks[b]
            if keep_ratio:
                gain = min(proc_h / h0, proc_w / w0)
                padw = round((proc_w - w0 * gain) / 2 - 0.1)
                padh = round((proc_h - h0 * gain) / 2 - 0.1)
                sh, sw = hm / proc_h, wm / proc_w
                m = m[:, int(max(padh, 0) * sh): int((proc_h - max(padh, 0)) * sh), int(max(padw, 0) * sw): int((proc_w - max(padw, 0)) * sw)]
            m = torch.nn.functional.interpolate(m.unsqueeze(0).float(), size=(h0, w0), mode="bilinear", align_corners=False).squeeze(0)
            out.append(m.clamp_(0, 1))
        return out

    def _preds_postprocess(self, outputs, processed_sizes, original_sizes, num_top_queries=300) -> List[Dict[str, torch.Tensor]]:
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        masks = outputs.get("pred_masks")
        B, Q, C = logits.shape
        full = self.process_boxes(boxes, processed_sizes, original_sizes, self.keep_ratio)
        k = min(num_top_queries, Q * C)
        labels, qidx, _, scores = kernels.detection_topk(logits, boxes.float(), k, int(processed_sizes[0][0]), int(processed_sizes[0][1]))
        top_boxes = full.gather(1, qidx.unsqueeze(-1).expand(-1, -1, 4))
        if self._thr is None or self._thr.device != scores.device:
            self._thr = torch.tensor(self.conf_threshs, device=scores.device, dtype=torch.float32)
        keep = scores >= self._thr[labels]                         # per-class thresholds, batch-wide
        results = []
        for b in range(B):
            kb = keep[b]
            out = {"labels": labels[b][kb], "boxes": top_boxes[b][kb], "scores": scores[b][kb]}
            if masks is not None and out["labels"].numel() > 0:
                mb = masks[b, qidx[b][kb]]
                m = self.process_masks(mb.unsqueeze(0), processed_sizes[b], [original_sizes[b]], self.keep_ratio)[0]
                if self.binarize_masks:
                    m = (m >= self.mask_threshold).to(torch.uint8)
                h, w = m.shape[-2:]
                ys = torch.arange(h, device=m.device)[None, :, None]
                xs = torch.arange(w, device=m.device)[None, None, :]
                x1, y1, x2, y2 = out["boxes"].T
                inside = (xs >= x1[:, None, None]) & (xs < x2[:, None, None]) & (ys >= y1[:, None, None]) & (ys < y2[:, None, None])
                out["masks"] = m * inside.to(m.dtype)
            results.append(out)
        return results

    @torch.no_grad()
    def _forward(self, inputs):
        if self.half and inputs.is_cuda:
            with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                return self.model(inputs)
        return self.model(inputs)

    @torch.no_grad()
    def _predict(self, inputs):
        if not (self.hip_graph and inputs.is_cuda):
            return self._forward(inputs)
        key = tuple(inputs.shape)
        ent = self._graphs.pop(key, None)
        if ent is None:
            # one captured graph (static input, outputs, private memory pool) per input shape: with `rect` inputs or varying
            # batch sizes the shapes are unbounded, so only the most recently used few are kept
            while len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._capture(inputs)
        self._graphs[key] = ent                            # (re-inserted: most recently used last)
        if ent is False:                                   # capture failed for this shape: eager from now on
            return self._forward(inputs)
        static_in, graph, static_out = ent
        static_in.copy_(inputs)
        graph.replay()
        # the graph's output tensors are overwritten by the next replay: hand out copies
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in static_out.items()} if isinstance(static_out, dict) else static_out

    def _capture(self, inputs):
        """Warm-up on a side stream (fills the anchor / packed-weight / constant caches), then one capture of the forward."""
        from .. import kernels as K
        static_in = inputs.clone()
        flags = (K._CAPTURE_POSSIBLE, K._CAPTURE_FROZEN_WEIGHTS)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._forward(static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            K._CAPTURE_POSSIBLE, K._CAPTURE_FROZEN_WEIGHTS = True, True
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward(static_in)
            return static_in, graph, out
        except Exception as e:                              # keep serving: eager path
            import warnings
            warnings.warn(f"Torch_model: HIP-graph capture failed for input {tuple(inputs.shape)} ({type(e).__name__}: {e}); running eagerly")
            torch.cuda.synchronize()
            return False
        finally:
            K._CAPTURE_POSSIBLE, K._CAPTURE_FROZEN_WEIGHTS = flags

    def _postprocess(self, preds, processed_sizes, original_sizes):
        output = self._preds_postprocess(preds, processed_sizes, original_sizes)
        if self.use_nms:
            for res in output:
                b, s, l, m = non_max_suppression(res["boxes"], res["scores"], res["labels"], res.get("masks"), 0.5)
                res["boxes"], res["scores"], res["labels"] = b, s, l
                if m is not None:
                    res["masks"] = m
        return output

    @torch.no_grad()
    def __call__(self, inputs) -> List[Dict[str, torch.Tensor]]:
        x, processed_sizes, original_sizes = self._prepare_inputs(inputs)
        return self._postprocess(self._predict(x), processed_sizes, original_sizes)
