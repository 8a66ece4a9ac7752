"""HungarianMatcher: bipartite matching between queries and ground-truth boxes.

Contract of the reference (`src/d_fine/matcher.py:74-257`): `matcher(outputs, targets)` returns
`{"indices": [(query_idx i64, target_idx i64)] * B}` as CPU tensors, query indices ascending,
exactly what `scipy.optimize.linear_sum_assignment` yields on the per-image cost block
    C = w_bbox * L1(cxcywh) + w_class * focal_cost + w_giou * (-GIoU)        (fp32, NaN -> 1)

MI355X design: only the block-diagonal [Q, T_i] costs are computed (the reference builds the
dense [B*Q, sum T] matrix), the assignment runs on the GPU (one workgroup per (head, image),
float64 shortest-augmenting-path with SciPy's tie-breaking), and `match_heads` solves all
L+2 prediction heads of a training step in ONE launch and ONE device->host copy instead of
L+2 blocking `.cpu()` round trips.
"""
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels


def dice_cost(pred_masks: torch.Tensor, gt_masks: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Pairwise 1 - Dice between [Q,H,W] probabilities and [T,H,W] masks (ref matcher.py:19-39)."""
    p = pred_masks.flatten(1).float()
    g = gt_masks.flatten(1).float()
    num = 2 * (p @ g.t())
    den = p.sum(1, keepdim=True) + g.sum(1)
    return 1 - (num + eps) / (den + eps)


def sigmoid_focal_cost(pred_logits, gt_labels, alpha: float = 0.25, gamma: float = 2.0):
    """Pairwise pixel-mean focal cost between [Q,HW] logits and [T,HW] masks (ref matcher.py:42-71)."""
    x = pred_logits.float()
    g = gt_labels.float()
    p = x.sigmoid()
    neg = (1 - alpha) * (p ** gamma) * (-(1 - p + 1e-8).log())
    pos = alpha * ((1 - p) ** gamma) * (-(p + 1e-8).log())
    return (pos @ g.t() + neg @ (1 - g).t()) / x.shape[1]


def _cols_to_pairs(cols: np.ndarray, sizes: List[int]):
    """target->query assignment vector(s) -> scipy-style (rows ascending, cols) per image."""
    out, off = [], 0
    for n in sizes:
        q = cols[off: off + n]
        t = np.nonzero(q >= 0)[0]
        order = np.argsort(q[t], kind="stable")
        out.append((torch.from_numpy(q[t][order].astype(np.int64)),
                    torch.from_numpy(t[order].astype(np.int64))))
        off += n
    return out


class Matching:
    """One head's matching of a whole batch as three flat int64 arrays (image, query, target-in-image),
    ordered like the reference's per-image (rows ascending, cols) pairs concatenated over the batch.
    Behaves like the reference's `indices` list ([(rows, cols)] * B of CPU int64 tensors) when it is
    indexed / iterated; the criterion's hot path reads the flat arrays and never builds that list."""

    __slots__ = ("img", "src", "tgt", "num_images", "_pairs")

    def __init__(self, img, src, tgt, num_images):
        self.img, self.src, self.tgt, self.num_images = img, src, tgt, num_images
        self._pairs = None

    @classmethod
    def from_pairs(cls, pairs):
        lens = [len(s) for s, _ in pairs]
        img = np.repeat(np.arange(len(pairs), dtype=np.int64), lens)
        src = np.concatenate([np.asarray(s, dtype=np.int64).reshape(-1) for s, _ in pairs]) if pairs else img
        tgt = np.concatenate([np.asarray(t, dtype=np.int64).reshape(-1) for _, t in pairs]) if pairs else img
        return cls(img, src, tgt, len(pairs))

    def pairs(self):
        if self._pairs is None:
            bounds = np.searchsorted(self.img, np.arange(self.num_images + 1))
            src, tgt = torch.from_numpy(self.src), torch.from_numpy(self.tgt)
            self._pairs = [(src[a:b], tgt[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
        return self._pairs

    def __len__(self):
        return self.num_images

    def __getitem__(self, i):
        return self.pairs()[i]

    def __iter__(self):
        return iter(self.pairs())


def _cols_to_matchings(cols: np.ndarray, sizes: List[int]):
    """[K, T] target->query vectors of K heads -> K `Matching`s, no per-image Python loop."""
    sizes = np.asarray(sizes, dtype=np.int64)
    nimg = len(sizes)
    img_of_t = np.repeat(np.arange(nimg, dtype=np.int64), sizes)
    t_local = np.arange(int(sizes.sum()), dtype=np.int64) - np.repeat(np.cumsum(sizes) - sizes, sizes)
    out = []
    for k in range(cols.shape[0]):
        q = cols[k].astype(np.int64)
        keep = np.nonzero(q >= 0)[0]
        img, src, tgt = img_of_t[keep], q[keep], t_local[keep]
        order = np.lexsort((src, img))                 # by image, then query ascending (queries are unique)
        out.append(Matching(img[order], src[order], tgt[order], nimg))
    return out


class HungarianMatcher(nn.Module):
    __share__ = ["use_focal_loss"]

    def __init__(self, weight_dict, use_focal_loss=False, alpha=0.25, gamma=2.0):
        super().__init__()
        self.cost_class = weight_dict["cost_class"]
        self.cost_bbox = weight_dict["cost_bbox"]
        self.cost_giou = weight_dict["cost_giou"]
        self.cost_mask = weight_dict.get("cost_mask", 0)
        self.cost_mask_dice = weight_dict.get("cost_mask_dice", 0)
        self.use_focal_loss, self.alpha, self.gamma = use_focal_loss, alpha, gamma
        assert self.cost_class != 0 or self.cost_bbox != 0 or self.cost_giou != 0, "all costs cant be 0"

    # ------------------------------------------------------------------ mask costs (segment task)
    def _mask_cost(self, outputs, targets, num_queries, tmax):
        """[B, Q, Tmax] extra cost from predicted masks, or None (ref matcher.py:175-237)."""
        if not (self.cost_mask > 0 or self.cost_mask_dice > 0) or outputs.get("pred_masks") is None:
            return None
        if not any(t.get("masks") is not None and t["masks"].numel() > 0 for t in targets):
            return None
        pm = outputs["pred_masks"]
        if pm.shape[1] != num_queries and pm.shape[1] > num_queries:
            pm = pm[:, pm.shape[1] - num_queries:]
        hm, wm = pm.shape[-2:]
        if pm.is_cuda and pm.dtype in (torch.float32, torch.bfloat16) and all(
                t.get("masks") is not None and len(t["masks"]) == len(t["boxes"]) for t in targets):
            return self._mask_cost_hip(pm, targets, num_queries, tmax)
        extra = torch.zeros(pm.shape[0], num_queries, tmax, device=pm.device)
        for b, t in enumerate(targets):
            n = len(t["boxes"])
            if n == 0 or t.get("masks") is None or t["masks"].numel() == 0:
                continue
            gt = t["masks"].float().to(pm.device)
            if gt.shape[-2:] != (hm, wm):
                gt = F.interpolate(gt.unsqueeze(1), size=(hm, wm), mode="bilinear",
                                   align_corners=False).squeeze(1)
            c = torch.zeros(num_queries, n, device=pm.device)
            if self.cost_mask_dice > 0:
                c = c + self.cost_mask_dice * dice_cost(pm[b].sigmoid(), gt)
            if self.cost_mask > 0:
                c = c + self.cost_mask * sigmoid_focal_cost(pm[b].flatten(1), gt.flatten(1),
                                                            alpha=self.alpha, gamma=self.gamma)
            extra[b, :, :n] = c
        return extra

    def gt_masks_at(self, targets, hm, wm, device):
        """The batch's target masks at mask resolution, concatenated [sum T, hm, wm] f32 (+ int32 offsets [B + 1] and the
        per-mask pixel sums): prepared once per step and shared by the cost kernel of every head and by the criterion's mask
        losses (the reference resizes them per image in every matcher call and every loss branch)."""
        # (identity of the list AND a fingerprint of its contents: a caller may refill or edit the same list object)
        key = (id(targets), hm, wm, str(device), tuple((t["masks"].data_ptr(), t["masks"]._version, len(t["boxes"]))
                                                        if t.get("masks") is not None else (0, 0, len(t["boxes"])) for t in targets))
        cache = getattr(self, "_gt_mask_cache", None)
        if cache is None or cache[0] != key:
            sizes = [len(t["boxes"]) for t in targets]
            chunks = []
            for t in targets:
                if len(t["boxes"]) == 0:
                    continue
                gt = t["masks"].float().to(device)
                if gt.shape[-2:] != (hm, wm):
                    gt = kernels.bilinear_resize(gt.unsqueeze(1), (hm, wm)).squeeze(1) if gt.is_cuda else \
                        F.interpolate(gt.unsqueeze(1), size=(hm, wm), mode="bilinear", align_corners=False).squeeze(1)
                chunks.append(gt)
            gt_all = torch.cat(chunks) if chunks else torch.zeros(0, hm, wm, device=device)
            toff = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), device=device)
            cache = self._gt_mask_cache = (key, gt_all.contiguous(), toff, gt_all.flatten(1).sum(1), targets)
        return cache[1], cache[2], cache[3]

    def _mask_cost_hip(self, pm, targets, num_queries, tmax, eps=1e-6):
        """The same costs from ONE pass over the mask logits per head (csrc/mask.hip: mask_cost_kernel accumulates, for every
        (query, target) of an image, sum_p sigmoid(x) g and sum_p (pos - neg)(x) g plus the per-query sums) instead of four
        [Q, HW] x [HW, T] matmuls on materialised sigmoid / focal maps per image."""
        hm, wm = pm.shape[-2:]
        sizes = [len(t["boxes"]) for t in targets]
        gt_all, toff, gsum = self.gt_masks_at(targets, hm, wm, pm.device)
        out, qsum = kernels.mask_cost_sums(pm, gt_all, toff, num_queries, tmax, self.alpha, self.gamma)
        tsum = torch.zeros(len(sizes), tmax, device=pm.device)
        col = torch.arange(tmax, device=pm.device)[None, :]
        valid = col < torch.as_tensor(sizes, device=pm.device)[:, None]
        tsum[valid] = gsum
        extra = torch.zeros(pm.shape[0], num_queries, tmax, device=pm.device)
        if self.cost_mask_dice > 0:
            extra = extra + self.cost_mask_dice * (1 - (2 * out[..., 0] + eps) / (qsum[:, :, None, 0] + tsum[:, None, :] + eps))
        if self.cost_mask > 0:
            extra = extra + self.cost_mask * (out[..., 1] + qsum[:, :, None, 1]) / float(hm * wm)
        return extra * valid[:, None, :]

    # ------------------------------------------------------------------ batched entry point
    @torch.no_grad()
    def match_heads(self, heads: List[Dict[str, torch.Tensor]], targets):
        """Matches every prediction head in `heads` (dicts with pred_logits [B,Q,C] and
        pred_boxes [B,Q,4]) against the same targets.  One cost+assignment launch and one
        D2H copy for all heads.  Returns one `Matching` (list-like: reference-style indices) per head."""
        return self.match_heads_async(heads, targets)()

    @torch.no_grad()
    def match_heads_async(self, heads: List[Dict[str, torch.Tensor]], targets):
        """Launches the matching and the (pinned, asynchronous) D2H copy of its result and returns a
        `finish()` callable that waits for it - the step's one host<->device sync - so that the caller
        can do host work that does not depend on the assignment while the device is still busy."""
        sizes = [len(t["boxes"]) for t in targets]
        tmax = max(sizes) if sizes else 0
        if tmax == 0:
            z = np.zeros(0, dtype=np.int64)
            empty = [Matching(z, z, z, len(sizes)) for _ in heads]
            return lambda: empty
        logits = torch.stack([h["pred_logits"] for h in heads]).float()
        boxes = torch.stack([h["pred_boxes"] for h in heads]).float()
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_box = torch.cat([t["boxes"] for t in targets]).float()
        extra = None
        per_head = [self._mask_cost(h, targets, logits.shape[2], tmax) for h in heads]
        if any(e is not None for e in per_head):
            extra = torch.stack([e if e is not None else torch.zeros_like(
                next(x for x in per_head if x is not None)) for e in per_head])
        cols, _ = kernels.hungarian_assign(
            logits, boxes, tgt_ids, tgt_box, sizes, float(self.cost_class), float(self.cost_bbox),
            float(self.cost_giou), float(self.alpha), float(self.gamma), self.use_focal_loss, extra)
        if not cols.is_cuda:
            return lambda: _cols_to_matchings(cols.numpy(), sizes)
        host = torch.empty(cols.shape, dtype=cols.dtype, pin_memory=True)
        host.copy_(cols, non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def finish():
            done.synchronize()               # the step's single host<->device sync
            return _cols_to_matchings(host.numpy(), sizes)

        return finish

    @torch.no_grad()
    def match_heads_device(self, heads: List[Dict[str, torch.Tensor]], targets):
        """The matching of every head left ON THE DEVICE: (cols int32 [K, T], tgt_offset int32 [B + 1], sizes) - `cols[k, t]`
        is the query assigned to target row t by head k - for the criterion's device-side bookkeeping (csrc/plans.hip), or
        None when that path does not apply (CPU tensors, no targets, an image with more targets than queries)."""
        sizes = [len(t["boxes"]) for t in targets]
        tmax = max(sizes) if sizes else 0
        first = heads[0]["pred_logits"]
        if tmax == 0 or not first.is_cuda:
            return None
        from .. import hip
        if not hip.criterion_plans_supported(len(heads), tmax, first.shape[1]):
            return None
        logits = torch.stack([h["pred_logits"] for h in heads]).float()
        boxes = torch.stack([h["pred_boxes"] for h in heads]).float()
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_box = torch.cat([t["boxes"] for t in targets]).float()
        extra = None
        per_head = [self._mask_cost(h, targets, logits.shape[2], tmax) for h in heads]
        if any(e is not None for e in per_head):
            extra = torch.stack([e if e is not None else torch.zeros_like(
                next(x for x in per_head if x is not None)) for e in per_head])
        cols, _ = kernels.hungarian_assign(
            logits, boxes, tgt_ids, tgt_box, sizes, float(self.cost_class), float(self.cost_bbox),
            float(self.cost_giou), float(self.alpha), float(self.gamma), self.use_focal_loss, extra)
        return cols, cols._dfine_tgt_offset, sizes

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets, return_topk=False):
        if return_topk:
            return {"indices_o2m": self.get_top_k_matches(outputs, targets, k=return_topk)}
        return {"indices": self.match_heads([outputs], targets)[0].pairs()}

    @torch.no_grad()
    def get_top_k_matches(self, outputs, targets, k=1):
        """k rounds of assignment; after each round the queries matched in an image are priced
        out (+inf -> FLT_MAX in the kernel) so the next round picks different queries.  The
        reference's variant (matcher.py:259-285, off the default path: the criterion never
        passes return_topk) writes 1e6 into `c[:, np.stack((rows, cols))]`, which prices out the
        union of matched query AND target indices in every image of the batch; that indexing
        quirk is not reproduced."""
        sizes = [len(t["boxes"]) for t in targets]
        logits = outputs["pred_logits"][None].float()
        boxes = outputs["pred_boxes"][None].float()
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_box = torch.cat([t["boxes"] for t in targets]).float()
        tmax = max(sizes)
        extra = torch.zeros(1, logits.shape[1], logits.shape[2], tmax, device=logits.device)
        rounds = []
        for _ in range(k):
            cols, _ = kernels.hungarian_assign(
                logits, boxes, tgt_ids, tgt_box, sizes, float(self.cost_class),
                float(self.cost_bbox), float(self.cost_giou), float(self.alpha), float(self.gamma),
                self.use_focal_loss, extra)
            pairs = _cols_to_pairs(cols.cpu().numpy()[0], sizes)
            rounds.append(pairs)
            for b, (q, _t) in enumerate(pairs):
                extra[0, b, q.to(extra.device)] = float("inf")
        return [(torch.cat([r[b][0] for r in rounds]), torch.cat([r[b][1] for r in rounds]))
                for b in range(len(sizes))]
