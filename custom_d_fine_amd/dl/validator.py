"""Detection metrics of the evaluation hand-off: precision / recall / F1 / mean IoU, TP / FP / FN counts, per-class values,
confusion matrix and COCO-style mAP for lists of per-image predictions and ground truth in the format
`dl/postprocess.py::preds_postprocess` / `gt_postprocess` produce.

Counterpart of the reference's `Validator` (`src/dl/validator.py:21-451`, box path):
  * `_compute_metrics_and_confusion_matrix` (`:354-451`): per image every (prediction, ground-truth) pair with IoU >= iou_thresh
    is visited in order of decreasing IoU and matched if both are still free; a matched pair of equal labels is a TP (its IoU
    is recorded), of different labels a FN for the GT class and a FP for the predicted class (IoU 0 recorded for both);
    unmatched predictions are FPs, unmatched ground truths FNs (IoU 0 each).  `_compute_main_metrics` (`:295-352`) sums them.
  * the reference delegates mAP to torchmetrics + faster_coco_eval (absent here): `coco_map` below is a plain restatement of
    the COCO protocol (IoU 0.50:0.05:0.95, 101-point interpolated precision, at most 100 detections per image, all areas) -
    **parity unpinned** against torchmetrics itself.
  * instance masks (`_compute_metrics_and_confusion_matrix_masks`, `:453-568`): when predictions AND ground truth carry masks
    the matching runs on the pairwise MASK IoU instead of the box IoU (predictions of another resolution are resized to the
    ground truth's with bilinear interpolation and re-thresholded at 0.5, float masks / `mask_probs` are binarised with
    `> conf_thresh`); `mAP_50_mask` / `mAP_50_95_mask` follow the same protocol as the box mAP on the mask IoU.  Device-resident
    masks are bit-packed and intersected with popcounts (`csrc/metrics.hip`, values bit-identical to the reference's fp32
    matmul route); the pycocotools RLE round trip the reference uses to keep a validation set in host memory
    (`src/dl/utils.py:1040-1160`) has no counterpart - masks stay on the device, one bit per pixel when packed.
The pairwise IoUs of all images are computed in ONE batched pass on the device the boxes live on (padded [images, P, G]); the
greedy matching - inherently sequential per image - runs on the host over the thresholded pairs."""
from collections import defaultdict
from typing import Dict, List

import numpy as np
import torch


def pairwise_iou_batched(pred_boxes: List[torch.Tensor], gt_boxes: List[torch.Tensor]) -> List[np.ndarray]:
    """Per image the [P_i, G_i] IoU matrix (xyxy), all images in one padded device pass."""
    n = len(pred_boxes)
    if n == 0:
        return []
    dev = pred_boxes[0].device
    pm = max([len(b) for b in pred_boxes] + [1])
    gm = max([len(b) for b in gt_boxes] + [1])
    P = torch.zeros(n, pm, 4, device=dev)
    G = torch.zeros(n, gm, 4, device=dev)
    for i, (p, g) in enumerate(zip(pred_boxes, gt_boxes)):
        if len(p):
            P[i, :len(p)] = p.to(dev, torch.float32)
        if len(g):
            G[i, :len(g)] = g.to(dev, torch.float32)
    area_p = (P[..., 2] - P[..., 0]) * (P[..., 3] - P[..., 1])
    area_g = (G[..., 2] - G[..., 0]) * (G[..., 3] - G[..., 1])
    lt = torch.maximum(P[:, :, None, :2], G[:, None, :, :2])
    rb = torch.minimum(P[:, :, None, 2:], G[:, None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_p[:, :, None] + area_g[:, None, :] - inter
    iou = (inter / union).cpu().numpy()
    return [iou[i, :len(p), :len(g)] for i, (p, g) in enumerate(zip(pred_boxes, gt_boxes))]


def _nhw(m):
    if m.ndim == 4 and m.shape[1] == 1:
        m = m[:, 0]
    return m


def binary_masks(sample, conf_thresh, keys=("masks", "mask_probs")):
    """[N, H, W] masks of a sample as the reference binarises them (validator.py:203-253): uint8 as is, anything else
    `> conf_thresh`; None when the sample has none."""
    for k in keys:
        m = sample.get(k)
        if m is not None and hasattr(m, "numel") and m.numel() > 0:
            m = _nhw(m)
            return m if m.dtype in (torch.uint8, torch.bool) else (m > float(conf_thresh)).to(torch.uint8)
    return None


def pairwise_mask_iou(pm: torch.Tensor, gm: torch.Tensor, thresh: float = 0.5) -> np.ndarray:
    """[Np, H, W] x [Ng, H, W] -> [Np, Ng] IoU (validator.py:283-293).  uint8 / bool masks count non-zero pixels, float masks
    pixels `> thresh`.  On the device: bit-packed masks + popcounts (csrc/metrics.hip); host tensors take the reference's
    composition."""
    if pm.shape[0] == 0 or gm.shape[0] == 0:
        return np.zeros((pm.shape[0], gm.shape[0]), dtype=np.float32)
    if pm.is_cuda:
        from .. import hip
        return hip.mask_iou_bits(hip.mask_pack_bits(pm, thresh), hip.mask_pack_bits(gm.to(pm.device), thresh)).cpu().numpy()

    def as01(m):
        return (m != 0 if m.dtype in (torch.uint8, torch.bool) else m > thresh).to(torch.float32).flatten(1)
    a, b = as01(pm), as01(gm)
    inter = a @ b.T
    union = a.sum(1, keepdim=True) + b.sum(1, keepdim=True).T - inter
    return torch.where(union > 0, inter / union, torch.zeros_like(union)).numpy()


class Validator:
    def __init__(self, gt: List[Dict[str, torch.Tensor]], preds: List[Dict[str, torch.Tensor]], label_to_name: Dict[int, str],
                 conf_thresh=0.5, iou_thresh=0.5, compute_maps=True) -> None:
        """gt[i] = {'labels' i64 [G], 'boxes' f32 [G, 4] absolute xyxy}; preds[i] = {'labels', 'boxes', 'scores'} already
        thresholded at conf_thresh, optionally 'all_labels' / 'all_boxes' / 'all_scores' (the unthresholded top-K) for mAP -
        the reference's format (validator.py:33-41,60-64)."""
        self.gt, self.preds = gt, preds
        self.conf_thresh, self.iou_thresh = conf_thresh, iou_thresh
        self.label_to_name = label_to_name
        self.compute_maps = compute_maps
        self.conf_matrix = None
        self.metrics_per_class = None
        self.class_to_idx = None
        # masks take part when both sides carry them (validator.py:69-79)
        self.use_masks = any(binary_masks(p, conf_thresh, ("masks",)) is not None for p in preds) and \
            any(binary_masks(g, conf_thresh, ("masks",)) is not None for g in gt)
        self._mask_ious = None

    def mask_ious(self):
        """Per image the [P, G] mask IoU matrix (predictions resized to the ground truth's resolution when they differ:
        bilinear, align_corners=False, > 0.5 - validator.py:485-491)."""
        if self._mask_ious is None:
            out = []
            for p, g in zip(self.preds, self.gt):
                pm, gm = binary_masks(p, self.conf_thresh), binary_masks(g, self.conf_thresh, ("masks",))
                n_p, n_g = len(p["labels"]), len(g["labels"])
                if pm is None or gm is None or n_p == 0 or n_g == 0:
                    out.append(np.zeros((n_p, n_g), dtype=np.float32))
                    continue
                if pm.shape[-2:] != gm.shape[-2:]:
                    if pm.is_cuda:
                        from .. import kernels
                        r = kernels.bilinear_resize(pm.unsqueeze(0).float().contiguous(), tuple(gm.shape[-2:]))[0]
                    else:
                        r = torch.nn.functional.interpolate(pm.unsqueeze(1).float(), size=gm.shape[-2:], mode="bilinear",
                                                            align_corners=False)[:, 0]
                    pm = (r > 0.5).to(torch.uint8)
                out.append(pairwise_mask_iou(pm, gm))
            self._mask_ious = out
        return self._mask_ious

    # ------------------------------------------------------------------------------------------------------------
    def _match(self, ignore_masks=False):
        per_class = defaultdict(lambda: {"TPs": 0, "FPs": 0, "FNs": 0, "IoUs": []})
        classes = set()
        pl = [p["labels"].cpu().numpy() for p in self.preds]
        gl = [g["labels"].cpu().numpy() for g in self.gt]
        for a in pl + gl:
            classes.update(a.tolist())
        classes = sorted(classes)
        idx = {c: i for i, c in enumerate(classes)}
        nc = len(classes)
        conf = np.zeros((nc + 1, nc + 1), dtype=int)
        if self.use_masks and not ignore_masks:
            ious = self.mask_ious()
        else:
            ious = pairwise_iou_batched([p["boxes"].reshape(-1, 4) for p in self.preds], [g["boxes"].reshape(-1, 4) for g in self.gt])
        for iou, plab, glab in zip(ious, pl, gl):
            n_p, n_g = len(plab), len(glab)
            used_p, used_g = np.zeros(n_p, bool), np.zeros(n_g, bool)
            if n_p and n_g:
                pi, gi = np.nonzero(iou >= self.iou_thresh)
                vals = iou[pi, gi]
                order = np.argsort(-vals, kind="stable")      # ties keep row-major (prediction, gt) order like torch.argsort of equal keys
                for k in order:
                    p, g = pi[k], gi[k]
                    if used_p[p] or used_g[g]:
                        continue
                    used_p[p] = used_g[g] = True
                    a, b = int(plab[p]), int(glab[g])
                    conf[idx[b], idx[a]] += 1
                    if a == b:
                        per_class[b]["TPs"] += 1
                        per_class[b]["IoUs"].append(float(vals[k]))
                    else:
                        per_class[b]["FNs"] += 1
                        per_class[a]["FPs"] += 1
                        per_class[b]["IoUs"].append(0)
                        per_class[a]["IoUs"].append(0)
            for p in np.nonzero(~used_p)[0]:
                a = int(plab[p])
                conf[nc, idx[a]] += 1
                per_class[a]["FPs"] += 1
                per_class[a]["IoUs"].append(0)
            for g in np.nonzero(~used_g)[0]:
                b = int(glab[g])
                conf[idx[b], nc] += 1
                per_class[b]["FNs"] += 1
                per_class[b]["IoUs"].append(0)
        return per_class, conf, idx

    def compute_metrics(self, extended=False, ignore_masks=False) -> Dict[str, float]:
        self.metrics_per_class, self.conf_matrix, self.class_to_idx = self._match(ignore_masks)
        tps = fps = fns = 0
        ious, ext = [], {}
        for key, v in self.metrics_per_class.items():
            tps, fps, fns = tps + v["TPs"], fps + v["FPs"], fns + v["FNs"]
            ious.extend(v["IoUs"])
            name = self.label_to_name[key]
            pr = v["TPs"] / (v["TPs"] + v["FPs"]) if v["TPs"] + v["FPs"] > 0 else 0
            rc = v["TPs"] / (v["TPs"] + v["FNs"]) if v["TPs"] + v["FNs"] > 0 else 0
            ext[f"precision_{name}"], ext[f"recall_{name}"] = pr, rc
            ext[f"iou_{name}"] = np.mean(v["IoUs"])
            ext[f"f1_{name}"] = 2 * pr * rc / (pr + rc) if pr + rc > 0 else 0
        precision = tps / (tps + fps) if tps + fps > 0 else 0
        recall = tps / (tps + fns) if tps + fns > 0 else 0
        out = {"f1": 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0,
               "precision": precision, "recall": recall, "iou": np.mean(ious) if ious else 0,
               "TPs": tps, "FPs": fps, "FNs": fns, "extended_metrics": ext}
        if self.compute_maps:
            m = coco_map(self.gt, self.preds)
            out["mAP_50"], out["mAP_50_95"] = m["map_50"], m["map"]
            if self.use_masks and not ignore_masks:            # validator.py:117-121 (segm mAP of the kept predictions' masks)
                mm = coco_map(self.gt, self.preds, ious=self.mask_ious())
                out["mAP_50_mask"], out["mAP_50_95_mask"] = mm["map_50"], mm["map"]
        if not extended:
            out.pop("extended_metrics", None)
        return out


def coco_map(gt, preds, max_dets=100, ious=None):
    """COCO detection mAP (all areas; `ious`: per image [P, G] IoU matrices of the thresholded predictions - the mask IoU of
    the segmentation mAP - instead of the box IoU of the `all_*` / plain fields): per class and IoU threshold t in 0.50:0.05:0.95 the detections of all images are sorted
    by score, each is matched to the still-free ground truth of its image and class with the highest IoU >= t, precision is made
    monotonically non-increasing and sampled at 101 recall points; classes without ground truth are skipped."""
    thr = np.arange(0.5, 0.96, 0.05)
    rec_pts = np.linspace(0.0, 1.0, 101)

    def field(p, k):
        return p[k] if ious is not None else p.get(f"all_{k}", p[k])

    if ious is None:
        pb = [field(p, "boxes").reshape(-1, 4) for p in preds]
        ious = pairwise_iou_batched(pb, [g["boxes"].reshape(-1, 4) for g in gt])
    pl = [field(p, "labels").cpu().numpy() for p in preds]
    ps = [field(p, "scores").cpu().numpy() for p in preds]
    gl = [g["labels"].cpu().numpy() for g in gt]
    classes = sorted(set(np.concatenate(gl).tolist())) if gl and sum(len(x) for x in gl) else []
    ap = np.full((len(thr), len(classes)), -1.0)
    for ci, c in enumerate(classes):
        dets = []                                            # (score, image, prediction index)
        n_gt = 0
        for i in range(len(gt)):
            n_gt += int((gl[i] == c).sum())
            sel = np.nonzero(pl[i] == c)[0]
            sel = sel[np.argsort(-ps[i][sel], kind="stable")][:max_dets]
            dets += [(ps[i][j], i, j) for j in sel]
        if n_gt == 0:
            continue
        dets.sort(key=lambda d: -d[0])
        for ti, t in enumerate(thr):
            taken = [np.zeros(len(g), bool) for g in gl]
            tp = np.zeros(len(dets), bool)
            for di, (_, i, j) in enumerate(dets):
                cand = np.nonzero((gl[i] == c) & ~taken[i])[0]
                if len(cand) == 0:
                    continue
                v = ious[i][j, cand]
                k = int(np.argmax(v))
                if v[k] >= t:
                    taken[i][cand[k]] = True
                    tp[di] = True
            ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
            recall = ctp / n_gt
            prec = ctp / np.maximum(ctp + cfp, 1e-12)
            for k in range(len(prec) - 1, 0, -1):
                prec[k - 1] = max(prec[k - 1], prec[k])
            inds = np.searchsorted(recall, rec_pts, side="left")
            q = np.zeros(101)
            ok = inds < len(prec)
            q[ok] = prec[inds[ok]]
            ap[ti, ci] = q.mean()
    valid = ap[ap > -1]
    return {"map": float(valid.mean()) if valid.size else -1.0,
            "map_50": float(ap[0][ap[0] > -1].mean()) if (ap.size and (ap[0] > -1).any()) else -1.0}
