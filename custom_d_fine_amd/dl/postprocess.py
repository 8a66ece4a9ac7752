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


def _resize(m: torch.Tensor, size) -> torch.Tensor:
    """[N, h, w] -> [N, H, W], bilinear, align_corners=False: the HIP gather kernel on the device (csrc/mask.hip)."""
    if m.is_cuda:
        return kernels.bilinear_resize(m.unsqueeze(0).contiguous(), (int(size[0]), int(size[1])))[0]
    return torch.nn.functional.interpolate(m.unsqueeze(0), size=(int(size[0]), int(size[1])), mode="bilinear", align_corners=False)[0]


def process_masks(pred_masks: torch.Tensor, processed_size, orig_sizes, keep_ratio: bool) -> List[torch.Tensor]:
    """[B, Q, Hm, Wm] (or [Q, Hm, Wm]) mask probabilities -> per image [Q, H0, W0] in [0, 1] in the ORIGINAL frame: resized to
    the network input size, the letterbox padding cut off (keep_ratio), resized to the original size (reference:
    src/dl/utils.py:715-769; fp32 throughout - the reference's evaluation loop detours through fp16 to save host memory)."""
    single = pred_masks.dim() == 3
    if single:
        pred_masks = pred_masks.unsqueeze(0)
    sizes = torch.as_tensor(orig_sizes).reshape(-1, 2).tolist()
    if pred_masks.shape[1] == 0:
        return [torch.zeros((0, int(sizes[0][0]), int(sizes[0][1])), device=pred_masks.device)]
    proc_h, proc_w = int(processed_size[0]), int(processed_size[1])
    out = []
    for b in range(pred_masks.shape[0]):
        h0, w0 = int(sizes[b][0]), int(sizes[b][1])
        m = _resize(pred_masks[b].float(), (proc_h, proc_w))
        if keep_ratio:
            gain = min(proc_h / h0, proc_w / w0)
            padw = round((proc_w - w0 * gain) / 2 - 0.1)
            padh = round((proc_h - h0 * gain) / 2 - 0.1)
            m = m[:, max(padh, 0): proc_h - max(padh, 0), max(padw, 0): proc_w - max(padw, 0)]
        out.append(_resize(m, (h0, w0)).clamp_(0, 1))
    return out


def cleanup_masks(masks: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    """Pixels outside the instance's box are cleared: x in [x1, x2), y in [y1, y2) on the integer pixel grid (utils.py:772-786)."""
    _, h, w = masks.shape
    ys = torch.arange(h, device=masks.device)[None, :, None]
    xs = torch.arange(w, device=masks.device)[None, None, :]
    x1, y1, x2, y2 = boxes.to(masks.device).T
    inside = (xs >= x1[:, None, None]) & (xs < x2[:, None, None]) & (ys >= y1[:, None, None]) & (ys < y2[:, None, None])
    return masks * inside.to(masks.dtype)


def preds_postprocess(inputs: torch.Tensor, outputs: Dict[str, torch.Tensor], orig_sizes, num_labels: int,
                      keep_ratio: bool, conf_thresh: float, num_top_queries: int = 300,
                      use_focal_loss: bool = True) -> List[Dict[str, torch.Tensor]]:
    """List (batch) of {"labels", "boxes", "scores", "all_boxes", "all_scores", "all_labels"} with the reference's
    meaning (train.py:240-332): top-K (query, class) pairs by sigmoid score, boxes mapped to the original frame,
    `labels/boxes/scores` thresholded at conf_thresh, `all_*` unthresholded.  Everything stays on the device until the
    final per-image split (the reference calls .cpu() six times per image).  With a mask head (`pred_masks` [B, Q, Hm, Wm]) the
    kept queries' masks are mapped to the original frame, binarised with >= conf_thresh and cleared outside their boxes
    (train.py:305-329); they STAY on the device as uint8 (the validator bit-packs them there - the reference moves them to the
    host and RLE-encodes them)."""
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    pred_masks = outputs.get("pred_masks")
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
        res = {"labels": labels_c[b][kb], "boxes": boxes_c[b][kb], "scores": scores_c[b][kb],
               "all_boxes": boxes_c[b], "all_scores": scores_c[b], "all_labels": labels_c[b]}
        if pred_masks is not None and int(kb.sum()) > 0:
            osz = torch.as_tensor(orig_sizes)[b].reshape(1, 2)
            probs = process_masks(pred_masks[b, qidx[b][keep[b]]].unsqueeze(0), inputs.shape[2:], osz, keep_ratio)[0]
            res["masks"] = cleanup_masks((probs >= conf_thresh).to(torch.uint8), top_boxes[b][keep[b]])
        results.append(res)
    return results


def gt_postprocess(inputs: torch.Tensor, targets, orig_sizes, keep_ratio: bool):
    """Ground truth in the same frame as `preds_postprocess` (reference train.py:334-378): {"labels", "boxes"[, "masks"]} per
    image; masks (network-size rasters) are mapped to the original frame and re-thresholded at 0.5, uint8 on their device."""
    out = []
    for t, osz in zip(targets, torch.as_tensor(orig_sizes).tolist()):
        bx = t["boxes"]
        if bx.numel():
            bx = process_boxes(bx[None], inputs.shape[2:], [osz], keep_ratio, bx.device)[0]
        res = {"labels": t["labels"].cpu(), "boxes": bx.cpu()}
        m = t.get("masks")
        if m is not None:
            if m.numel() > 0:
                pm = process_masks(m.to(inputs.device, torch.float32).unsqueeze(0), inputs.shape[2:], [osz], keep_ratio)[0]
                res["masks"] = (pm >= 0.5).to(torch.uint8)
            else:
                res["masks"] = torch.zeros((0, int(osz[0]), int(osz[1])), dtype=torch.uint8, device=inputs.device)
        out.append(res)
    return out
