"""Evaluation-side post-processing of the decoder outputs, batch-wide on the device.

Counterpart of the reference's `Trainer.preds_postprocess` / `gt_postprocess` (`src/dl/train.py:240-365`) and the box
mapping helpers in `src/dl/utils.py:160-185,624-712` (`norm_xywh_to_abs_xyxy`, `scale_boxes`, `scale_boxes_ratio_kept`,
`process_boxes`).  The reference moves boxes to numpy and loops over the images on the host; here the mapping is a
handful of broadcast tensor ops on the device the predictions live on, and the sigmoid/top-K/label split is the HIP
post-processor kernel (`kernels.detection_topk`, csrc/postproc.hip).  Values follow the reference's fp32 arithmetic
operation by operation (per-image scalars are computed in float64 like Python floats, then applied in fp32).
"""
from typing import Dict, List

import torch

from .. import kernels


def _sizes(orig_sizes, device):
    t = torch.as_tensor(orig_sizes, device=device)
    return t.to(torch.float64)


def process_boxes(boxes: torch.Tensor, processed_size, orig_sizes, keep_ratio: bool, device=None) -> torch.Tensor:
    """boxes [B, N, 4] normalised cxcywh (network-input frame) -> absolute xyxy in the ORIGINAL image frame.
    processed_size (h, w) of the network input; orig_sizes [B, 2] (h, w).  Reference: src/dl/utils.py:673-712."""
    device = boxes.device if device is None else torch.device(device)
    # the reference multiplies its float32 numpy boxes by numpy INT64 sizes (not weak Python scalars), which promotes the
    # whole conversion to float64; the rounded corners then land in a float32 array (utils.py:160-177,690-697)
    b = boxes.to(device=device, dtype=torch.float64)
    ph, pw = float(processed_size[0]), float(processed_size[1])
    xc, yc, bw, bh = b[..., 0] * pw, b[..., 1] * ph, b[..., 2] * pw, b[..., 3] * ph
    x0 = torch.clamp(torch.floor(xc - bw / 2), min=1).float()
    y0 = torch.clamp(torch.floor(yc - bh / 2), min=1).float()
    x1 = torch.clamp(torch.ceil(xc + bw / 2), max=pw - 1).float()
    y1 = torch.clamp(torch.ceil(yc + bh / 2), max=ph - 1).float()
    orig = _sizes(orig_sizes, device)                       # [B, 2] float64 (h, w)
    oh, ow = orig[:, 0:1], orig[:, 1:2]
    if keep_ratio:
        gain = torch.minimum(ph / oh, pw / ow)
        padw = torch.round((pw - ow * gain) / 2 - 0.1)
        padh = torch.round((ph - oh * gain) / 2 - 0.1)
        # float32 array -= int (exact), then /= float64 gain: divided in float64, stored as float32 (utils.py:651-653)
        x0, x1 = ((x0 - padw.float()).double() / gain).float(), ((x1 - padw.float()).double() / gain).float()
        y0, y1 = ((y0 - padh.float()).double() / gain).float(), ((y1 - padh.float()).double() / gain).float()
        ow32, oh32 = ow.float(), oh.float()
        zero = torch.zeros_like(ow32)
        x0, x1 = torch.maximum(torch.minimum(x0, ow32), zero), torch.maximum(torch.minimum(x1, ow32), zero)
        y0, y1 = torch.maximum(torch.minimum(y0, oh32), zero), torch.maximum(torch.minimum(y1, oh32), zero)
    else:
        sx, sy = ow / pw, oh / ph                           # float64 scalars applied to the float32 corners (utils.py:664-670)
        x0, x1 = (x0.double() * sx).float(), (x1.double() * sx).float()
        y0, y1 = (y0.double() * sy).float(), (y1.double() * sy).float()
    return torch.stack([x0, y0, x1, y1], dim=-1)


def preds_postprocess(inputs: torch.Tensor, outputs: Dict[str, torch.Tensor], orig_sizes, num_labels: int,
                      keep_ratio: bool, conf_thresh: float, num_top_queries: int = 300,
                      use_focal_loss: bool = True) -> List[Dict[str, torch.Tensor]]:
    """List (batch) of {"labels", "boxes", "scores", "all_boxes", "all_scores", "all_labels"} with the reference's
    meaning (train.py:240-332): top-K (query, class) pairs by sigmoid score, boxes mapped to the original frame,
    `labels/boxes/scores` thresholded at conf_thresh, `all_*` unthresholded.  Everything stays on the device until the
    final per-image split (the reference calls .cpu() six times per image)."""
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    B, Q, C = logits.shape
    if not use_focal_loss:
        raise NotImplementedError("softmax scoring is off the default path (configs.py: use_focal_loss=True)")
    full = process_boxes(boxes, inputs.shape[2:], orig_sizes, keep_ratio, inputs.device)        # [B, Q, 4]
    k = min(num_top_queries, Q * C)
    labels, qidx, _, scores = kernels.detection_topk(logits, boxes, k, int(inputs.shape[2]), int(inputs.shape[3]))
    top_boxes = full.gather(1, qidx.unsqueeze(-1).expand(-1, -1, 4))
    keep = scores >= conf_thresh
    labels_c, boxes_c, scores_c, keep_c = labels.cpu(), top_boxes.cpu(), scores.cpu(), keep.cpu()   # one hop per field
    results = []
    for b in range(B):
        kb = keep_c[b]
        results.append({"labels": labels_c[b][kb], "boxes": boxes_c[b][kb], "scores": scores_c[b][kb],
                        "all_boxes": boxes_c[b], "all_scores": scores_c[b], "all_labels": labels_c[b]})
    return results


def gt_postprocess(inputs: torch.Tensor, targets, orig_sizes, keep_ratio: bool):
    """Ground truth in the same frame as `preds_postprocess` (reference train.py:334-365): {"labels", "boxes"} per image."""
    out = []
    for t, osz in zip(targets, torch.as_tensor(orig_sizes).tolist()):
        bx = t["boxes"]
        if bx.numel():
            bx = process_boxes(bx[None], inputs.shape[2:], [osz], keep_ratio, bx.device)[0]
        out.append({"labels": t["labels"].cpu(), "boxes": bx.cpu()})
    return out
