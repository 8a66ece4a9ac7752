"""DFINECriterion: VFL + L1/GIoU + FGL/DDF (+ cropped mask BCE/Dice) over the main, auxiliary,
pre, encoder and denoising heads.

Same constructor, `forward(outputs, targets) -> dict[str, 0-d tensor]`, loss names and
denominators as the reference (`src/d_fine/dfine_criterion.py`).  Restructured for the GPU:
  * all L+2 Hungarian matchings of a step run in one launch (`matcher.match_heads`);
  * matched (image, query, target) triples are uploaded once per index set and reused by
    every loss of every head instead of being rebuilt from python lists per loss;
  * no `.item()` / `torch.equal` / `mask.any()` host syncs inside the loss loops (the only
    device->host copy of the step is the matcher's assignment vector).
"""
import copy

import numpy as np
import torch
import torch.distributed
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels
from .arch.utils import upload, bbox2distance, box_cxcywh_to_xyxy, box_iou, generalized_box_iou, paired_iou_giou
from .dist_utils import get_world_size, is_dist_available_and_initialized


import os

_DEVICE_PLANS = [os.environ.get("DFINE_DEVICE_PLANS", "1") == "1"]      # a list: tools / tests flip it in place


def _as_matching(indices):
    from .matcher import Matching
    return indices if isinstance(indices, Matching) else Matching.from_pairs(indices)


class _Plan:
    """Device-side gather plan of one matching: which (image, query) pairs are matched to
    which row of the batch-concatenated targets, packed as int64 [3, M]."""

    def __init__(self, indices, offsets, device, packed=None):
        m = _as_matching(indices)
        if packed is None:
            packed = upload(self.pack(m, offsets), device)
        self.packed = packed
        self.count = int(m.src.size)
        self.count_dev = None                # (a plan built from host indices: its length is a host number)
        self.indices = indices

    # rows of the packed plan (only the torch-composition path reads them)
    @property
    def batch(self):
        return self.packed[0]

    @property
    def src(self):
        return self.packed[1]

    @property
    def tgt(self):
        return self.packed[2]

    @staticmethod
    def pack(m, offsets):
        offs = np.asarray(offsets[:-1], dtype=np.int64)
        return np.stack([m.img, m.src, m.tgt + (offs[m.img] if m.img.size else m.img)])


class _DevPlan:
    """A gather plan that was built on the device (csrc/plans.hip): int64 [3, capacity] rows (image, query, target row);
    `count` is the host's knowledge of its length (exact for a head's own matching: every target is matched; the
    capacity for the GO union, whose true length lives in `count_dev`)."""

    def __init__(self, packed, count, count_dev=None):
        self.packed, self.count, self.count_dev = packed, int(count), count_dev

    @property
    def batch(self):
        return self.packed[0]

    @property
    def src(self):
        return self.packed[1]

    @property
    def tgt(self):
        return self.packed[2]


class DFINECriterion(nn.Module):
    __share__ = ["num_classes"]
    __inject__ = ["matcher"]

    def __init__(self, matcher, weight_dict, losses, alpha=0.2, gamma=2.0, num_classes=80,
                 reg_max=32, boxes_weight_format=None, share_matched_indices=False,
                 label_smoothing: float = 0.0):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.losses = losses
        self.boxes_weight_format = boxes_weight_format
        self.share_matched_indices = share_matched_indices
        self.alpha, self.gamma, self.reg_max = alpha, gamma, reg_max
        self.label_smoothing = label_smoothing
        self._clear_cache()
        self._tgt = None

    # ------------------------------------------------------------------ bookkeeping
    def _clear_cache(self):
        self.fgl_targets, self.fgl_targets_dn = None, None
        self.own_targets, self.own_targets_dn = None, None
        self.num_pos, self.num_neg = None, None
        self._plans = {}

    def _targets_cat(self, targets):
        """(labels [T], boxes [T,4], per-image offsets) of the batch, concatenated once."""
        key = targets           # the list itself: the cache keeps it alive, so its identity cannot be recycled by another list
        if self._tgt is None or self._tgt[0] is not key:
            sizes = [len(t["labels"]) for t in targets]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + n)
            self._tgt = (key, torch.cat([t["labels"] for t in targets]),
                         torch.cat([t["boxes"] for t in targets]), offs)
        return self._tgt[1], self._tgt[2], self._tgt[3]

    def _plan(self, indices, targets, device) -> _Plan:
        if isinstance(indices, (_DevPlan, _Plan)):
            return indices
        key = id(indices)
        if key not in self._plans:
            self._plans[key] = _Plan(indices, self._targets_cat(targets)[2], device)
        return self._plans[key]

    def _build_plans(self, index_sets, targets, device):
        """All gather plans of a step in ONE host->device copy (each pageable H2D copy stalls the
        launch queue for ~0.1 ms)."""
        offsets = self._targets_cat(targets)[2]
        todo = [ix for ix in index_sets if id(ix) not in self._plans]
        if not todo:
            return
        packs = [_Plan.pack(_as_matching(ix), offsets) for ix in todo]
        flat = upload(np.concatenate([p.reshape(-1) for p in packs]), device)
        off = 0
        for ix, p in zip(todo, packs):
            n = p.size
            self._plans[id(ix)] = _Plan(ix, offsets, device, packed=flat[off: off + n].view(3, -1))
            off += n

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for src, _ in indices])

    def _get_tgt_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for _, tgt in indices])

    def _matched_boxes(self, outputs, targets, indices):
        p = self._plan(indices, targets, outputs["pred_boxes"].device)
        src = outputs["pred_boxes"][p.batch, p.src]
        tgt = self._targets_cat(targets)[1][p.tgt]
        return p, src, tgt

    # ------------------------------------------------------------------ classification
    def _class_targets(self, src_logits, p: _Plan, labels):
        cls = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                         device=src_logits.device)
        cls[p.batch, p.src] = labels[p.tgt]
        return cls

    def loss_labels_focal(self, outputs, targets, indices, num_boxes):
        """Sigmoid focal loss on one-hot (optionally smoothed) targets (ref :67-90)."""
        x = outputs["pred_logits"]
        p = self._plan(indices, targets, x.device)
        cls = self._class_targets(x, p, self._targets_cat(targets)[0])
        tgt = F.one_hot(cls, num_classes=self.num_classes + 1)[..., :-1].float()
        if self.label_smoothing is not None and self.label_smoothing > 0:
            tgt = tgt * (1 - self.label_smoothing) + self.label_smoothing / tgt.shape[-1]
        prob = torch.sigmoid(x)
        ce = F.binary_cross_entropy_with_logits(x, tgt, reduction="none")
        p_t = prob * tgt + (1 - prob) * (1 - tgt)
        loss = ce * ((1 - p_t) ** self.gamma)
        if self.alpha >= 0:
            loss = (self.alpha * tgt + (1 - self.alpha) * (1 - tgt)) * loss
        return {"loss_focal": loss.mean(1).sum() * x.shape[1] / num_boxes}

    def loss_labels_vfl(self, outputs, targets, indices, num_boxes, values=None):
        """Varifocal loss: BCE against IoU-valued soft labels, negatives down-weighted by
        alpha * p^gamma (ref dfine_criterion.py:92-122)."""
        x = outputs["pred_logits"]
        p, src, tgt = self._matched_boxes(outputs, targets, indices)
        if values is None:
            ious, _ = paired_iou_giou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt))
            ious = ious.detach()
        else:
            ious = values
        cls = self._class_targets(x, p, self._targets_cat(targets)[0])
        onehot = F.one_hot(cls, num_classes=self.num_classes + 1)[..., :-1]
        score = torch.zeros(x.shape[:2], dtype=x.dtype, device=x.device)
        score[p.batch, p.src] = ious.to(x.dtype)
        tgt_score = score.unsqueeze(-1) * onehot
        prob = torch.sigmoid(x).detach()
        weight = self.alpha * prob.pow(self.gamma) * (1 - onehot) + tgt_score
        loss = F.binary_cross_entropy_with_logits(x, tgt_score, weight=weight, reduction="none")
        return {"loss_vfl": loss.mean(1).sum() * x.shape[1] / num_boxes}

    # ------------------------------------------------------------------ boxes
    def loss_boxes(self, outputs, targets, indices, num_boxes, boxes_weight=None):
        """L1 + (1 - GIoU) over matched pairs (ref dfine_criterion.py:124-143)."""
        _, src, tgt = self._matched_boxes(outputs, targets, indices)
        sx, tx = box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)
        if not sx.is_cuda:  # the reference asserts on degenerate boxes (arch/utils.py:41-42); on the
            assert (sx[:, 2:] >= sx[:, :2]).all()  # GPU that check would be a host sync per loss call
        _, giou = paired_iou_giou(sx, tx)
        g = 1 - giou
        if boxes_weight is not None:
            g = g * boxes_weight
        return {"loss_bbox": F.l1_loss(src, tgt, reduction="none").sum() / num_boxes,
                "loss_giou": g.sum() / num_boxes}

    # ------------------------------------------------------------------ FGL + DDF
    def loss_local(self, outputs, targets, indices, num_boxes, T=5):
        """Fine-grained localisation loss on the matched edge distributions and decoupled
        distillation (KL to the last layer's distributions) on all of them
        (ref dfine_criterion.py:145-237)."""
        losses = {}
        if "pred_corners" not in outputs:
            return losses
        nb = self.reg_max + 1
        is_dn = "is_dn" in outputs
        p, src, tgt = self._matched_boxes(outputs, targets, indices)
        corners_all = outputs["pred_corners"]
        pred = corners_all[p.batch, p.src].reshape(-1, nb)
        ref = outputs["ref_points"][p.batch, p.src].detach()
        tgt_xyxy = box_cxcywh_to_xyxy(tgt)
        with torch.no_grad():
            if is_dn and self.fgl_targets_dn is None:
                self.fgl_targets_dn = bbox2distance(ref, tgt_xyxy, self.reg_max,
                                                    outputs["reg_scale"], outputs["up"])
            if not is_dn and self.fgl_targets is None:
                self.fgl_targets = bbox2distance(ref, tgt_xyxy, self.reg_max,
                                                 outputs["reg_scale"], outputs["up"])
        t_corner, w_right, w_left = self.fgl_targets_dn if is_dn else self.fgl_targets

        ious, _ = paired_iou_giou(box_cxcywh_to_xyxy(src), tgt_xyxy)
        w_iou = ious.unsqueeze(-1).repeat(1, 1, 4).reshape(-1).detach()
        losses["loss_fgl"] = self.unimodal_distribution_focal_loss(
            pred, t_corner, w_right, w_left, w_iou, avg_factor=num_boxes)

        if "teacher_corners" in outputs:
            teacher = outputs["teacher_corners"]
            if teacher is corners_all or (teacher.data_ptr() == corners_all.data_ptr()
                                          and teacher.shape == corners_all.shape):
                # the teacher itself (last dn layer): KL(p||p) == 0, as the reference's
                # torch.equal branch (ref :197-198) - decided without a device sync
                losses["loss_ddf"] = corners_all.sum() * 0
            else:
                b, q = corners_all.shape[:2]
                w_loc = outputs["teacher_logits"].sigmoid().max(dim=-1)[0].detach().clone()
                matched = torch.zeros(b, q, dtype=torch.bool, device=w_loc.device)
                matched[p.batch, p.src] = True
                w_loc[p.batch, p.src] = ious.detach().to(w_loc.dtype)
                w_loc = w_loc.unsqueeze(-1).repeat(1, 1, 4).reshape(-1)
                mask = matched.unsqueeze(-1).repeat(1, 1, 4).reshape(-1)
                kl = F.kl_div(F.log_softmax(corners_all.reshape(-1, nb) / T, dim=1),
                              F.softmax(teacher.reshape(-1, nb).detach() / T, dim=1),
                              reduction="none").sum(-1)
                per_row = w_loc * (T ** 2) * kl
                n_pos = mask.sum()
                n_neg = mask.numel() - n_pos
                if not is_dn:
                    scale = 8 / b  # keeps the pos/neg balance independent of the per-GPU batch
                    self.num_pos, self.num_neg = (n_pos * scale) ** 0.5, (n_neg * scale) ** 0.5
                fm = mask.to(per_row.dtype)
                l_pos = (per_row * fm).sum() / n_pos.clamp(min=1)
                l_neg = (per_row * (1 - fm)).sum() / n_neg.clamp(min=1)
                losses["loss_ddf"] = (l_pos * self.num_pos + l_neg * self.num_neg) / (
                    self.num_pos + self.num_neg)
        return losses

    def unimodal_distribution_focal_loss(self, pred, label, weight_right, weight_left,
                                         weight=None, reduction="sum", avg_factor=None):
        """Two-bin cross entropy against the left/right bins of a continuous target."""
        left = label.long()
        logp = F.log_softmax(pred, dim=1)
        loss = -(logp.gather(1, left[:, None]).squeeze(1) * weight_left.reshape(-1)
                 + logp.gather(1, left[:, None] + 1).squeeze(1) * weight_right.reshape(-1))
        if weight is not None:
            loss = loss * weight.float()
        if avg_factor is not None:
            return loss.sum() / avg_factor
        return loss.mean() if reduction == "mean" else loss.sum()

    # ------------------------------------------------------------------ masks (segment task)
    def _prepare_target_masks(self, targets, indices, out_h, out_w, device):
        chunks = []
        for t, (_, j) in zip(targets, indices):
            m = t.get("masks")
            if m is None or m.numel() == 0 or m.dim() != 3 or j.numel() == 0:
                continue
            sel = m[j.to(m.device)].float().to(device)
            sel = kernels.bilinear_resize(sel.unsqueeze(1), (out_h, out_w)) if sel.is_cuda else \
                F.interpolate(sel.unsqueeze(1), size=(out_h, out_w), mode="bilinear", align_corners=False)
            chunks.append(sel.squeeze(1).clamp_(0, 1))
        if not chunks:
            return torch.zeros(0, out_h, out_w, device=device, dtype=torch.float32), 0
        out = torch.cat(chunks, dim=0)
        return out, out.shape[0]

    def _prepare_target_boxes_for_masks(self, targets, indices, out_h, out_w, device):
        chunks = []
        for t, (_, j) in zip(targets, indices):
            m = t.get("masks")
            if m is None or m.numel() == 0 or m.dim() != 3 or j.numel() == 0:
                continue
            b = t["boxes"][j.to(t["boxes"].device)]
            cx, cy, w, h = b.unbind(1)
            chunks.append(torch.stack([
                ((cx - w / 2) * out_w).clamp(0, out_w - 1), ((cy - h / 2) * out_h).clamp(0, out_h - 1),
                ((cx + w / 2) * out_w).clamp(1, out_w), ((cy + h / 2) * out_h).clamp(1, out_h)],
                dim=1).to(device))
        if not chunks:
            return torch.zeros(0, 4, device=device, dtype=torch.float32)
        return torch.cat(chunks, dim=0)

    @staticmethod
    def _inside(boxes, h, w, device, dtype):
        ys = torch.arange(h, device=device, dtype=dtype)[None, :, None]
        xs = torch.arange(w, device=device, dtype=dtype)[None, None, :]
        x1, y1, x2, y2 = (boxes[:, i:i + 1, None] for i in range(4))
        return ((xs >= x1) & (xs < x2)).float() * ((ys >= y1) & (ys < y2)).float(), (x1, y1, x2, y2)

    @staticmethod
    def _cropped_bce_loss(pred_logits, tgt_masks, boxes, eps=1e-6):
        """BCE inside the GT box only, normalised by the box area (ref :335-386)."""
        if pred_logits.shape[0] == 0:
            return pred_logits.sum() * 0.0
        _, h, w = pred_logits.shape
        inside, (x1, y1, x2, y2) = DFINECriterion._inside(boxes, h, w, pred_logits.device,
                                                          pred_logits.dtype)
        bce = F.binary_cross_entropy_with_logits(pred_logits, tgt_masks, reduction="none") * inside
        area = ((x2 - x1).reshape(-1) * (y2 - y1).reshape(-1)).clamp(min=1.0)
        return (bce.sum(dim=(1, 2)) / area).mean()

    @staticmethod
    def _cropped_dice_loss(pred_logits, tgt_masks, boxes, eps=1e-6):
        """Dice inside the GT box only (ref :404-450)."""
        if pred_logits.shape[0] == 0:
            return pred_logits.sum() * 0.0
        _, h, w = pred_logits.shape
        inside, _ = DFINECriterion._inside(boxes, h, w, pred_logits.device, pred_logits.dtype)
        p = (pred_logits.sigmoid() * inside).flatten(1)
        t = (tgt_masks * inside).flatten(1)
        dice = 1.0 - (2.0 * (p * t).sum(1) + eps) / (p.sum(1) + t.sum(1) + eps)
        return dice.mean()

    @staticmethod
    def _dice_loss(pred_logits, tgt_masks, eps=1e-6):
        p = pred_logits.sigmoid().flatten(1)
        t = tgt_masks.flatten(1)
        dice = 1.0 - (2.0 * (p * t).sum(1) + eps) / (p.sum(1) + t.sum(1) + eps)
        return dice.mean() if dice.numel() > 0 else pred_logits.sum() * 0.0

    def loss_masks(self, outputs, targets, indices, num_boxes):
        if "pred_masks" not in outputs:
            return {}
        pm = outputs["pred_masks"]
        _, _, hm, wm = pm.shape
        p = self._plan(indices, targets, pm.device)
        if p.count == 0:
            z = pm.sum() * 0
            return {"loss_mask_bce": z, "loss_mask_dice": z}
        if (pm.is_cuda and pm.dtype in (torch.float32, torch.bfloat16) and hasattr(self.matcher, "gt_masks_at") and all(
                t.get("masks") is not None and t["masks"].dim() == 3 and len(t["masks"]) == len(t["boxes"]) for t in targets)):
            # fused: targets of the whole batch at mask resolution once per step (shared with the matcher); the matched
            # prediction planes, target planes and boxes are read in place through the plan - no per-image host loop
            gt_all, _, _ = self.matcher.gt_masks_at(targets, hm, wm, pm.device)
            bkey = (id(targets), hm, wm, tuple((t["boxes"].data_ptr(), t["boxes"]._version, len(t["boxes"])) for t in targets))
            if getattr(self, "_mask_box_cache", (None,))[0] != bkey:
                b = self._targets_cat(targets)[1].float().to(pm.device)
                cx, cy, w, h = b.unbind(1)
                self._mask_box_cache = (bkey, torch.stack([
                    ((cx - w / 2) * wm).clamp(0, wm - 1), ((cy - h / 2) * hm).clamp(0, hm - 1),
                    ((cx + w / 2) * wm).clamp(1, wm), ((cy + h / 2) * hm).clamp(1, hm)], dim=1).contiguous(), targets)
            bce, dice = kernels.mask_losses(pm, p.batch, p.src, gt_all, self._mask_box_cache[1], plan_t=p.tgt)
            return {"loss_mask_bce": bce, "loss_mask_dice": dice}
        tgt, valid = self._prepare_target_masks(targets, indices, hm, wm, device=pm.device)
        if valid == 0:
            z = pm.sum() * 0
            return {"loss_mask_bce": z, "loss_mask_dice": z}
        boxes = self._prepare_target_boxes_for_masks(targets, indices, hm, wm, device=pm.device)
        sel = pm[p.batch, p.src]
        if sel.shape[0] != tgt.shape[0]:
            raise AssertionError(f"Mismatch between number of selected predictions ({sel.shape[0]})"
                                 f"and target masks ({tgt.shape[0]})")
        return {"loss_mask_bce": self._cropped_bce_loss(sel, tgt, boxes),
                "loss_mask_dice": self._cropped_dice_loss(sel, tgt, boxes)}

    # ------------------------------------------------------------------ GO indices
    def _get_go_indices(self, indices, indices_aux_list):
        """Union of the matchings of all heads; a query matched to different targets keeps the
        target it was matched to most often (ref dfine_criterion.py:570-591).  The reference does this
        per image with torch.unique(dim=0) (lexicographic pairs) + torch.argsort(counts, descending)
        + first-seen-per-query.  CPU argsort is not stable, so ties are broken by whatever ATen's sort
        does: the one thing kept per image is that very call on the same counts vector; the unique /
        first-seen bookkeeping around it is batch-wide numpy."""
        from .matcher import Matching
        sets = [_as_matching(indices)] + [_as_matching(a) for a in indices_aux_list]
        nimg = sets[0].num_images
        img = np.concatenate([m.img for m in sets])
        src = np.concatenate([m.src for m in sets])
        tgt = np.concatenate([m.tgt for m in sets])
        if img.size == 0:
            return Matching(img, src, tgt, nimg)
        qs, ts = int(src.max()) + 1, int(tgt.max()) + 1
        key, counts = np.unique((img * qs + src) * ts + tgt, return_counts=True)   # lexicographic (img, q, t)
        iq, t = key // ts, key % ts                           # iq = image * qs + query
        bounds = np.searchsorted(iq // qs, np.arange(nimg + 1))
        counts_t = torch.from_numpy(counts)
        perm = [a + torch.argsort(counts_t[a:b], descending=True).numpy()
                for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist()) if b > a]
        perm = np.concatenate(perm)
        iq, t = iq[perm], t[perm]
        _, first = np.unique(iq, return_index=True)           # first appearance of every (image, query)
        first.sort()                                          # ... in order of appearance
        iq, t = iq[first], t[first]
        return Matching(iq // qs, iq % qs, t, nimg)

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {"boxes": self.loss_boxes, "focal": self.loss_labels_focal,
                 "vfl": self.loss_labels_vfl, "local": self.loss_local, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    def _branch(self, outputs, targets, suffix, pick, losses_out, only_boxes_go=False):
        """All configured losses of one prediction head; `pick(loss)` -> (indices, num_boxes)."""
        for loss in self.losses:
            idx, nb = pick(loss)
            meta = self.get_loss_meta_info(loss, outputs, targets, idx)
            ld = self.get_loss(loss, outputs, targets, idx, nb, **meta)
            for k, v in ld.items():
                if k in self.weight_dict:
                    losses_out[k + suffix] = v * self.weight_dict[k]

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _upcast(obj, memo):
        """fp32 view of a (nested) head dict.  The model may run under bf16 autocast; the loss is
        always evaluated in fp32 (reference train.py:572-573 disables autocast around the loss).
        Tensor identity is preserved through `memo` (the DDF teacher check relies on it)."""
        if isinstance(obj, torch.Tensor):
            if not obj.dtype.is_floating_point or obj.dtype == torch.float32:
                return obj
            if id(obj) not in memo:
                memo[id(obj)] = obj.float()
            return memo[id(obj)]
        if isinstance(obj, dict):
            return {k: (v if k in ("dn_meta", "enc_meta") else DFINECriterion._upcast(v, memo))
                    for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(DFINECriterion._upcast(v, memo) for v in obj)
        return obj

    def forward(self, outputs, targets, **kwargs):
        assert "aux_outputs" in outputs, ""
        device = outputs["pred_logits"].device
        fused = device.type == "cuda" and self._fusable(outputs)
        if not fused:
            outputs = self._upcast(outputs, {})
        main = {k: v for k, v in outputs.items() if "aux" not in k}
        heads = [main] + list(outputs["aux_outputs"]) + [outputs["pre_outputs"]] + list(
            outputs["enc_aux_outputs"])
        self._clear_cache()
        self._tgt = None
        indices_dn = None
        if fused and _DEVICE_PLANS[0] and hasattr(self.matcher, "match_heads_device"):
            # no host <-> device synchronisation: the matching stays on the device, the gather plans, the GO union and the
            # normalisers that depend on its size are built there (csrc/plans.hip)
            dev_match = self.matcher.match_heads_device(heads, targets)
            if dev_match is not None:
                return self._forward_fused_device(outputs, targets, heads, *dev_match)
        if hasattr(self.matcher, "match_heads_async"):
            finish = self.matcher.match_heads_async(heads, targets)
            # host / launch work that does not need the assignment runs while the device computes it
            self._targets_cat(targets)
            if fused:
                self._fdr_constants(outputs)
                if "dn_outputs" in outputs:
                    indices_dn = self.get_cdn_matched_indices(outputs["dn_meta"], targets)
            matched = finish()
        elif hasattr(self.matcher, "match_heads"):
            matched = self.matcher.match_heads(heads, targets)
        else:
            matched = [self.matcher(h, targets)["indices"] for h in heads]
        n_aux = len(outputs["aux_outputs"])
        indices = matched[0]
        cached = matched[1: n_aux + 2]          # aux layers ... pre
        cached_enc = matched[n_aux + 2:]
        indices_go = self._get_go_indices(indices, matched[1:])

        # the reference's two scalar all-reduces (dfine_criterion.py:639-652) folded into one 2-float host-side collective
        from .dist_utils import host_all_reduce_sum
        world = get_world_size()
        n_tgt = float(sum(len(t["labels"]) for t in targets))
        if world > 1 and fused and _DEVICE_PLANS[0] and hasattr(self.matcher, "match_heads_device"):
            # Another rank of this step may be on the device-plan path (it is chosen per rank from the rank's OWN batch: no
            # targets, too many pairs for the plan kernel ...).  Every rank must issue the SAME collectives in the same order
            # or they pair up with the wrong partner (a gradient bucket's all-reduce): one 1-float device all-reduce of the
            # GO count, then one 1-float host all-reduce of the target count - exactly `_forward_fused_device`'s sequence.
            go_f = torch.full((1,), float(indices_go.src.size), device=device, dtype=torch.float32)
            torch.distributed.all_reduce(go_f)
            tot = [float(go_f.item()), host_all_reduce_sum([n_tgt])[0]]
        else:
            tot = host_all_reduce_sum([float(indices_go.src.size), n_tgt])
        # fp32 division like the reference's torch.clamp(t / world, min=1).item()
        num_boxes_go = max(float(np.float32(tot[0]) / np.float32(world)), 1.0)
        num_boxes = max(float(np.float32(tot[1]) / np.float32(world)), 1.0)
        # what the losses were normalised by (tests/test_dist_gpu.py checks them against the reference's definition)
        self.__dict__["_last_norm"] = {"num_boxes": num_boxes, "num_boxes_go": num_boxes_go, "go_count": int(indices_go.src.size)}

        def go_or(own, go_for=("boxes", "local")):
            return lambda loss: (indices_go, num_boxes_go) if loss in go_for else (own, num_boxes)

        if fused:
            return self._forward_fused(outputs, targets, indices, cached, cached_enc, indices_go,
                                       num_boxes, num_boxes_go, indices_dn)

        losses = {}
        self._branch(outputs, targets, "", go_or(indices), losses)

        for i, aux in enumerate(outputs["aux_outputs"]):
            aux["up"], aux["reg_scale"] = outputs["up"], outputs["reg_scale"]
            self._branch(aux, targets, f"_aux_{i}", go_or(cached[i]), losses)

        self._branch(outputs["pre_outputs"], targets, "_pre", go_or(cached[-1]), losses)

        assert "enc_meta" in outputs, ""
        agnostic = outputs["enc_meta"]["class_agnostic"]
        enc_targets = targets
        if agnostic:
            orig_nc, self.num_classes = self.num_classes, 1
            enc_targets = copy.deepcopy(targets)
            for t in enc_targets:
                t["labels"] = torch.zeros_like(t["labels"])
            self._tgt = None
        for i, aux in enumerate(outputs["enc_aux_outputs"]):
            self._branch(aux, enc_targets, f"_enc_{i}", go_or(cached_enc[i], go_for=("boxes",)), losses)
        if agnostic:
            self.num_classes = orig_nc
            self._tgt = None

        if "dn_outputs" in outputs:
            assert "dn_meta" in outputs, ""
            indices_dn = self.get_cdn_matched_indices(outputs["dn_meta"], targets)
            dn_boxes = num_boxes * outputs["dn_meta"]["dn_num_group"]
            dn_boxes = dn_boxes if dn_boxes > 0 else 1
            for i, aux in enumerate(outputs["dn_outputs"]):
                aux["is_dn"] = True
                aux["up"], aux["reg_scale"] = outputs["up"], outputs["reg_scale"]
                self._branch(aux, targets, f"_dn_{i}", lambda loss: (indices_dn, dn_boxes), losses)
            if "dn_pred_masks" in outputs and "masks" in self.losses:
                final = {"pred_masks": outputs["dn_pred_masks"],
                         "pred_boxes": outputs["dn_outputs"][-1]["pred_boxes"]}
                for k, v in self.loss_masks(final, targets, indices_dn, dn_boxes).items():
                    if k in self.weight_dict:
                        losses[k + "_dn_final"] = v * self.weight_dict[k]
            if "dn_pre_outputs" in outputs:
                self._branch(outputs["dn_pre_outputs"], targets, "_dn_pre",
                             lambda loss: (indices_dn, dn_boxes), losses)

        return {k: torch.nan_to_num(v, nan=0.0) for k, v in losses.items()}

    # ------------------------------------------------------------------ fused GPU path
    def _fusable(self, outputs):
        """The HIP head-loss kernels cover the reference's default configuration."""
        return (set(self.losses) <= {"vfl", "boxes", "local", "masks"} and self.boxes_weight_format is None
                and self.reg_max == 32 and not outputs["enc_meta"]["class_agnostic"])

    def _fdr_constants(self, outputs):
        """W(n) table and reg_scale as host numbers (model constants; fetched once)."""
        key = (id(outputs["up"]), id(outputs["reg_scale"]))
        if getattr(self, "_fdr_cache", (None,))[0] != key:
            from .arch.utils import weighting_function
            w = weighting_function(self.reg_max, outputs["up"].detach().float(),
                                   outputs["reg_scale"].detach().float()).cpu().tolist()
            self._fdr_cache = (key, w, float(outputs["reg_scale"].detach().float().cpu()))
        return self._fdr_cache[1], self._fdr_cache[2]

    def _forward_fused(self, outputs, targets, indices, cached, cached_enc, indices_go, num_boxes,
                       num_boxes_go, indices_dn=None):
        from .. import kernels
        dev = outputs["pred_logits"].device
        labels, tboxes, _ = self._targets_cat(targets)
        labels = labels.to(torch.int64)
        tboxes = tboxes.float().contiguous()
        wd = self.weight_dict
        wtable, reg_scale = self._fdr_constants(outputs)
        want_vfl, want_box, want_local = ("vfl" in self.losses, "boxes" in self.losses, "local" in self.losses)
        if indices_dn is None and "dn_outputs" in outputs:
            indices_dn = self.get_cdn_matched_indices(outputs["dn_meta"], targets)
        self._build_plans([indices, *cached, *cached_enc, indices_go] + ([indices_dn] if indices_dn is not None else []),
                          targets, dev)
        names, vecs = [], []
        want_masks, extra = "masks" in self.losses, {}

        def run(head, suffix, cls_idx, box_idx, n_cls, n_box, local, is_dn=False, box_go_only=False):
            cls_plan = self._plan(cls_idx, targets, dev)
            box_plan = self._plan(box_idx, targets, dev)
            b, q = head["pred_logits"].shape[:2]
            corners = head.get("pred_corners") if (local and want_local) else None
            teacher = head.get("teacher_corners") if corners is not None else None
            c_pos = c_neg = 0.0
            if teacher is not None:
                if teacher is corners or (teacher.data_ptr() == corners.data_ptr() and teacher.shape == corners.shape):
                    teacher = None          # the teacher itself: KL == 0 (ref dfine_criterion.py:197-198)
                else:
                    rows_pos = 4.0 * box_plan.count
                    rows_neg = 4.0 * (b * q) - rows_pos
                    if not is_dn:           # cached for the dn heads like the reference (:223-229)
                        scale = 8.0 / b
                        self.num_pos, self.num_neg = (rows_pos * scale) ** 0.5, (rows_neg * scale) ** 0.5
                    den = self.num_pos + self.num_neg
                    c_pos = wd["loss_ddf"] * self.num_pos / (den * rows_pos) if rows_pos > 0 else 0.0
                    c_neg = wd["loss_ddf"] * self.num_neg / (den * rows_neg) if rows_neg > 0 else 0.0
            cfg = {"wtable": wtable, "reg_max": self.reg_max, "reg_scale": reg_scale, "alpha": self.alpha,
                   "gamma": self.gamma, "temp": 5.0,
                   "s_vfl": wd["loss_vfl"] / n_cls if want_vfl else 0.0,
                   "s_l1": wd["loss_bbox"] / n_box if want_box else 0.0,
                   "s_giou": wd["loss_giou"] / n_box if want_box else 0.0,
                   "s_fgl": wd["loss_fgl"] / n_box, "c_pos": c_pos, "c_neg": c_neg}
            if q == 0:
                # denoising heads of a batch without targets ([B, 0, C]): the reference's terms are means / sums over nothing,
                # NaN -> 0 by its nan_to_num (dfine_criterion.py:776) - the keys exist, the values are zero
                vec = torch.zeros(5, device=dev, dtype=torch.float32)
            else:
                vec = kernels.head_losses(
                    head["pred_logits"], head["pred_boxes"], corners,
                    head["ref_points"].detach() if corners is not None else None, teacher,
                    head.get("teacher_logits") if teacher is not None else None, cls_plan.packed,
                    box_plan.packed, labels, tboxes, cfg)
            keys = []
            if want_vfl:
                keys.append(("loss_vfl", 0))
            if want_box:
                keys += [("loss_bbox", 1), ("loss_giou", 2)]
            if corners is not None:
                keys.append(("loss_fgl", 3))
                if "teacher_corners" in head:
                    keys.append(("loss_ddf", 4))
            vecs.append(vec)
            names.append([(k + suffix, j) for k, j in keys])
            if want_masks and head.get("pred_masks") is not None and q > 0:
                # the matched mask planes are read in place in the model's dtype (no fp32 copy of [B, Q, H/4, W/4] per head)
                for k, v in self.loss_masks(head, targets, cls_idx, n_cls).items():
                    if k in wd:
                        extra[k + suffix] = torch.nan_to_num(v * wd[k], nan=0.0)

        run(outputs, "", indices, indices_go, num_boxes, num_boxes_go, True)
        for i, aux in enumerate(outputs["aux_outputs"]):
            run(aux, f"_aux_{i}", cached[i], indices_go, num_boxes, num_boxes_go, True)
        run(outputs["pre_outputs"], "_pre", cached[-1], indices_go, num_boxes, num_boxes_go, False)
        for i, aux in enumerate(outputs["enc_aux_outputs"]):
            run(aux, f"_enc_{i}", cached_enc[i], indices_go, num_boxes, num_boxes_go, False)
        if indices_dn is not None:
            dn_boxes = num_boxes * outputs["dn_meta"]["dn_num_group"]
            dn_boxes = dn_boxes if dn_boxes > 0 else 1
            for i, aux in enumerate(outputs["dn_outputs"]):
                run(aux, f"_dn_{i}", indices_dn, indices_dn, dn_boxes, dn_boxes, True, is_dn=True)
            if want_masks and outputs.get("dn_pred_masks") is not None:
                final = {"pred_masks": outputs["dn_pred_masks"], "pred_boxes": outputs["dn_outputs"][-1]["pred_boxes"]}
                for k, v in self.loss_masks(final, targets, indices_dn, dn_boxes).items():
                    if k in wd:
                        extra[k + "_dn_final"] = torch.nan_to_num(v * wd[k], nan=0.0)
            if "dn_pre_outputs" in outputs:
                run(outputs["dn_pre_outputs"], "_dn_pre", indices_dn, indices_dn, dn_boxes, dn_boxes, False,
                    is_dn=True)
        table = torch.nan_to_num(torch.stack(vecs), nan=0.0)      # [heads, 5]
        self.__dict__["_last_table"] = table      # plain attribute: nn.Module.__setattr__ is not needed here
        cells = table.view(-1).unbind(0)                            # one op instead of ~60 selects
        losses = {}
        for h, keys in enumerate(names):
            for k, j in keys:
                losses[k] = cells[h * 5 + j]
        self.__dict__["_last_extra"] = list(extra.values())
        losses.update(extra)
        return losses

    def _forward_fused_device(self, outputs, targets, heads, cols, tgt_offset, sizes):
        """The fused head losses with every index set and every size-dependent scalar produced on the device: same launches as
        `_forward_fused`, fed with device-built plans (`_DevPlan`) and a device-resident table of the scalar factors."""
        from .. import hip, kernels
        from .dist_utils import host_all_reduce_sum
        dev = outputs["pred_logits"].device
        labels, tboxes, _ = self._targets_cat(targets)
        labels = labels.to(torch.int64)
        tboxes = tboxes.float().contiguous()
        wd = self.weight_dict
        wtable, reg_scale = self._fdr_constants(outputs)
        want_vfl, want_box, want_local = ("vfl" in self.losses, "boxes" in self.losses, "local" in self.losses)
        want_masks, extra = "masks" in self.losses, {}
        world = get_world_size()
        T = int(sum(sizes))
        q_main = outputs["pred_logits"].shape[1]
        head_plans, go_packed, go_count, go_f = hip.criterion_plans(cols, tgt_offset, sizes, q_main, want_float_count=world > 1)
        go_sum = None
        if world > 1:
            torch.distributed.all_reduce(go_f)        # tiny device collective on the stream (the reference: all_reduce + .item())
            go_sum = go_f
        tot = host_all_reduce_sum([float(T)])
        num_boxes = max(float(np.float32(tot[0]) / np.float32(world)), 1.0)
        # (the GO normaliser clamp(go_sum / world, 1) is formed on the device by dfine_criterion_scales; go_count stays there)
        self.__dict__["_last_norm"] = {"num_boxes": num_boxes, "num_boxes_go": None, "go_count": go_count}
        n_aux = len(outputs["aux_outputs"])
        plans = [_DevPlan(head_plans[k], T) for k in range(len(heads))]
        go = _DevPlan(go_packed, go_packed.shape[1], go_count)
        indices_dn = dn_plan = None
        dn_boxes = 1
        if "dn_outputs" in outputs:
            indices_dn = self.get_cdn_matched_indices(outputs["dn_meta"], targets)
            self._build_plans([indices_dn], targets, dev)
            dn_plan = self._plan(indices_dn, targets, dev)
            dn_boxes = num_boxes * outputs["dn_meta"]["dn_num_group"]
            dn_boxes = dn_boxes if dn_boxes > 0 else 1

        # ---- pass 1: the launches of the step and the host-known part of their scalar factors
        runs = []

        def add(head, suffix, cls_plan, box_plan, n_cls, n_box, local, is_dn=False):
            b, q = head["pred_logits"].shape[:2]
            corners = head.get("pred_corners") if (local and want_local) else None
            teacher = head.get("teacher_corners") if corners is not None else None
            if teacher is not None and (teacher is corners or (teacher.data_ptr() == corners.data_ptr() and teacher.shape == corners.shape)):
                teacher = None          # the teacher itself: KL == 0 (ref dfine_criterion.py:197-198)
            uses_go = box_plan is go
            row = (wd["loss_vfl"] / n_cls if want_vfl else 0.0, 1.0 if uses_go else 0.0, float(n_box),
                   wd["loss_bbox"] if want_box else 0.0, wd["loss_giou"] if want_box else 0.0, wd["loss_fgl"],
                   wd.get("loss_ddf", 0.0), 1.0 if teacher is not None else 0.0, 1.0 if is_dn else 0.0, 4.0 * (b * q),
                   4.0 * box_plan.count, 8.0 / b)
            runs.append((head, suffix, cls_plan, box_plan, corners, teacher, n_cls, row))

        add(outputs, "", plans[0], go, num_boxes, 1.0, True)
        for i, aux in enumerate(outputs["aux_outputs"]):
            add(aux, f"_aux_{i}", plans[1 + i], go, num_boxes, 1.0, True)
        add(outputs["pre_outputs"], "_pre", plans[n_aux + 1], go, num_boxes, 1.0, False)
        for i, aux in enumerate(outputs["enc_aux_outputs"]):
            add(aux, f"_enc_{i}", plans[n_aux + 2 + i], go, num_boxes, 1.0, False)
        if dn_plan is not None:
            for i, aux in enumerate(outputs["dn_outputs"]):
                add(aux, f"_dn_{i}", dn_plan, dn_plan, dn_boxes, dn_boxes, True, is_dn=True)
            if "dn_pre_outputs" in outputs:
                add(outputs["dn_pre_outputs"], "_dn_pre", dn_plan, dn_plan, dn_boxes, dn_boxes, False, is_dn=True)
        params = upload(np.asarray([r[-1] for r in runs], dtype=np.float64), dev)
        scales = hip.criterion_scales(params, go_count, go_sum, world)

        # ---- pass 2: the head-loss launches.  Their packed zero-initialised output blocks are slices of ONE arena: one fill per
        # step instead of one per head
        names, vecs = [], []
        sizes_z = [hip.head_losses_zbytes(h["pred_logits"].shape[0], h["pred_logits"].shape[1],
                                          c.shape[-1] if c is not None else 0, h["pred_logits"].element_size())
                   for h, _, _, _, c, *_ in runs]
        arena = torch.zeros(sum(sizes_z), device=dev, dtype=torch.uint8) if kernels.HEAD_ARENA else None
        z_off = 0
        for r, (head, suffix, cls_plan, box_plan, corners, teacher, n_cls, _) in enumerate(runs):
            cfg = {"wtable": wtable, "reg_max": self.reg_max, "reg_scale": reg_scale, "alpha": self.alpha, "gamma": self.gamma,
                   "temp": 5.0, "s_vfl": 0.0, "s_l1": 0.0, "s_giou": 0.0, "s_fgl": 0.0, "c_pos": 0.0, "c_neg": 0.0,
                   "scales_dev": scales[r], "box_count_dev": box_plan.count_dev,
                   "zbytes": None if arena is None else arena[z_off:z_off + sizes_z[r]]}
            z_off += sizes_z[r]
            vec = kernels.head_losses(
                head["pred_logits"], head["pred_boxes"], corners,
                head["ref_points"].detach() if corners is not None else None, teacher,
                head.get("teacher_logits") if teacher is not None else None, cls_plan.packed,
                box_plan.packed, labels, tboxes, cfg)
            keys = []
            if want_vfl:
                keys.append(("loss_vfl", 0))
            if want_box:
                keys += [("loss_bbox", 1), ("loss_giou", 2)]
            if corners is not None:
                keys.append(("loss_fgl", 3))
                if "teacher_corners" in head:
                    keys.append(("loss_ddf", 4))
            vecs.append(vec)
            names.append([(k + suffix, j) for k, j in keys])
            if want_masks and head.get("pred_masks") is not None:
                for k, v in self.loss_masks(head, targets, cls_plan, n_cls).items():
                    if k in wd:
                        extra[k + suffix] = torch.nan_to_num(v * wd[k], nan=0.0)
        if dn_plan is not None and want_masks and outputs.get("dn_pred_masks") is not None:
            final = {"pred_masks": outputs["dn_pred_masks"], "pred_boxes": outputs["dn_outputs"][-1]["pred_boxes"]}
            for k, v in self.loss_masks(final, targets, dn_plan, dn_boxes).items():
                if k in wd:
                    extra[k + "_dn_final"] = torch.nan_to_num(v * wd[k], nan=0.0)
        table = torch.nan_to_num(torch.stack(vecs), nan=0.0)      # [heads, 5]
        self.__dict__["_last_table"] = table
        cells = table.view(-1).unbind(0)
        losses = {}
        for h, keys in enumerate(names):
            for k, j in keys:
                losses[k] = cells[h * 5 + j]
        self.__dict__["_last_extra"] = list(extra.values())
        losses.update(extra)
        return losses

    def total(self, loss_dict):
        """Sum of all returned losses.  On the fused path the values are views of one table, so
        the sum is one reduction instead of len(loss_dict) - 1 scalar adds."""
        t = self.__dict__.get("_last_table")
        # the table is dropped here: holding it until the next step would keep that step's whole autograd
        # graph alive and free it (~1 ms of node destructors) in the middle of the next criterion call,
        # while the device idles
        self.__dict__["_last_table"] = None
        extra = self.__dict__.pop("_last_extra", None) or []
        if t is not None and loss_dict and next(iter(loss_dict.values()))._base is t:
            return t.sum() + sum(extra) if extra else t.sum()
        return sum(loss_dict.values())

    def get_loss_meta_info(self, loss, outputs, targets, indices):
        if self.boxes_weight_format is None:
            return {}
        _, src, tgt = self._matched_boxes(outputs, targets, indices)
        iou, giou = paired_iou_giou(box_cxcywh_to_xyxy(src.detach()), box_cxcywh_to_xyxy(tgt))
        if self.boxes_weight_format == "iou":
            val = iou
        elif self.boxes_weight_format == "giou":
            val = giou
        else:
            raise AttributeError()
        if loss in ("boxes",):
            return {"boxes_weight": val}
        if loss in ("vfl",):
            return {"values": val}
        return {}

    @staticmethod
    def get_cdn_matched_indices(dn_meta, targets):
        """Denoising queries are matched to their source GT by construction
        (ref dfine_criterion.py:809-831): positive slot g * 2 * gmax + j of image i <-> its GT j, for every
        group g.  Built batch-wide from the per-image GT counts (list-like result, see matcher.Matching)."""
        from .matcher import Matching
        pos, groups = dn_meta["dn_positive_idx"], dn_meta["dn_num_group"]
        counts = np.asarray([len(t["labels"]) for t in targets], dtype=np.int64)
        if pos is None:                 # a batch without targets: the reference's meta carries None (arch/utils.py:371-374)
            assert int(counts.sum()) == 0
            pos = []
        for i, n in enumerate(counts):
            assert n == 0 or len(pos[i]) == n * groups
        flat = dn_meta.get("dn_positive_flat")
        if flat is None:
            src = (np.concatenate([np.asarray(p, dtype=np.int64).reshape(-1) for p in pos])
                   if len(pos) else np.zeros(0, np.int64))
        else:
            src = flat
        img = np.repeat(np.arange(len(counts), dtype=np.int64), counts * groups)
        tgt = (np.concatenate([np.tile(np.arange(n, dtype=np.int64), groups) for n in counts])
               if len(counts) else np.zeros(0, np.int64))
        return Matching(img, src, tgt, len(counts))

    def feature_loss_function(self, fea, target_fea):
        loss = (fea - target_fea) ** 2 * ((fea > 0) | (target_fea > 0)).float()
        return torch.abs(loss)

    def get_gradual_steps(self, outputs):
        n = len(outputs["aux_outputs"]) + 1 if "aux_outputs" in outputs else 1
        return [0.5 + 0.5 / (n - 1) * i for i in range(n)] if n > 1 else [1]
