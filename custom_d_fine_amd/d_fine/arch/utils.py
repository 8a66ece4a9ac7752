"""Box algebra, FDR helpers, denoising-group builder and the deformable-attention entry point.

Public names and call contracts follow the reference's `src/d_fine/arch/utils.py`; the
arithmetic-heavy ones dispatch to the HIP library through `custom_d_fine_amd.kernels`.
"""
import math
from typing import List

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from ... import kernels


# ------------------------------------------------------------------ boxes
def box_area(b: Tensor) -> Tensor:
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(boxes1: Tensor, boxes2: Tensor):
    """Pairwise IoU and union of xyxy boxes -> ([N,M], [N,M]).  (ref arch/utils.py:12-26)"""
    a1, a2 = box_area(boxes1), box_area(boxes2)
    tl = torch.maximum(boxes1[:, None, :2], boxes2[None, :, :2])
    br = torch.minimum(boxes1[:, None, 2:], boxes2[None, :, 2:])
    ext = (br - tl).clamp(min=0)
    inter = ext[..., 0] * ext[..., 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Pairwise GIoU of xyxy boxes; asserts on degenerate boxes like the reference
    (arch/utils.py:29-51)."""
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    tl = torch.minimum(boxes1[:, None, :2], boxes2[None, :, :2])
    br = torch.maximum(boxes1[:, None, 2:], boxes2[None, :, 2:])
    ext = (br - tl).clamp(min=0)
    hull = ext[..., 0] * ext[..., 1]
    return iou - (hull - union) / hull


def paired_iou_giou(src_xyxy: Tensor, tgt_xyxy: Tensor):
    """Row-wise (diagonal) IoU and GIoU of two [M,4] xyxy sets.  The reference builds the
    full M x M matrix and takes `torch.diag` (dfine_criterion.py:98-99,137-139); only the
    diagonal is ever used, so it is computed directly with the same operation order."""
    a1 = (src_xyxy[:, 2] - src_xyxy[:, 0]) * (src_xyxy[:, 3] - src_xyxy[:, 1])
    a2 = (tgt_xyxy[:, 2] - tgt_xyxy[:, 0]) * (tgt_xyxy[:, 3] - tgt_xyxy[:, 1])
    tl = torch.maximum(src_xyxy[:, :2], tgt_xyxy[:, :2])
    br = torch.minimum(src_xyxy[:, 2:], tgt_xyxy[:, 2:])
    ext = (br - tl).clamp(min=0)
    inter = ext[:, 0] * ext[:, 1]
    union = a1 + a2 - inter
    iou = inter / union
    tl2 = torch.minimum(src_xyxy[:, :2], tgt_xyxy[:, :2])
    br2 = torch.maximum(src_xyxy[:, 2:], tgt_xyxy[:, 2:])
    ext2 = (br2 - tl2).clamp(min=0)
    hull = ext2[:, 0] * ext2[:, 1]
    return iou, iou - (hull - union) / hull


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    x = x.clip(min=0.0, max=1.0)
    return torch.log(x.clip(min=eps) / (1 - x).clip(min=eps))


def box_cxcywh_to_xyxy(x: Tensor) -> Tensor:
    cx, cy, w, h = x.unbind(-1)
    hw, hh = 0.5 * w.clamp(min=0.0), 0.5 * h.clamp(min=0.0)
    return torch.stack([cx - hw, cy - hh, cx + hw, cy + hh], dim=-1)


def box_xyxy_to_cxcywh(x: Tensor) -> Tensor:
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def bias_init_with_prob(prior_prob=0.01):
    return float(-math.log((1 - prior_prob) / prior_prob))


_ACTS = {
    "silu": nn.SiLU, "swish": nn.SiLU, "relu": nn.ReLU, "leaky_relu": nn.LeakyReLU,
    "gelu": nn.GELU, "hardsigmoid": nn.Hardsigmoid,
}


def get_activation(act, inpace: bool = True):
    if act is None:
        return nn.Identity()
    if isinstance(act, nn.Module):
        return act
    key = act.lower()
    if key not in _ACTS:
        raise RuntimeError("")
    m = _ACTS[key]()
    if hasattr(m, "inplace"):
        m.inplace = inpace
    return m


# ------------------------------------------------------------------ FDR
def weighting_function(reg_max, up, reg_scale, deploy=False):
    """Non-uniform bin positions W(n), n = 0..reg_max  (ref arch/utils.py:145-188).

    W = [-2b, -(s^(h-1))+1, ..., -(s^1)+1, 0, s^1-1, ..., s^(h-1)-1, 2b] with
    b = |up|*|reg_scale|, h = reg_max/2, s = (b+1)^(2/(reg_max-2)).
    """
    # `up` / `reg_scale` are constants of the model (parameters with requires_grad=False): the table - ~80 one-element
    # kernels - is rebuilt only when they change, not on every forward / criterion call
    cacheable = (not deploy and torch.is_tensor(up) and torch.is_tensor(reg_scale) and not up.requires_grad
                 and not reg_scale.requires_grad)
    if cacheable:
        # keyed on the tensor OBJECTS (held by the entry, so their ids / storage cannot be recycled by another model) and their
        # version counters; writers that bypass the counter (`.data` writes: FusedAdamWEMA.broadcast_from_rank0) call
        # invalidate_weighting_cache()
        key = (int(reg_max), id(up), up._version, id(reg_scale), reg_scale._version, up.data_ptr(), reg_scale.data_ptr(),
               up.dtype, str(up.device))
        hit = _W_CACHE.get(key)
        if hit is not None and hit[0] is up and hit[1] is reg_scale:
            return hit[2]
    w = _weighting_function(reg_max, up, reg_scale, deploy)
    if cacheable:
        if len(_W_CACHE) > 16:
            _W_CACHE.clear()
        _W_CACHE[key] = (up, reg_scale, w)
    return w


_W_CACHE = {}


def invalidate_weighting_cache():
    """Forget the cached W(n) tables (after `up` / `reg_scale` were rewritten through `.data`)."""
    _W_CACHE.clear()


def _weighting_function(reg_max, up, reg_scale, deploy):
    b1 = abs(up[0]) * abs(reg_scale)
    b2 = b1 * 2
    half = reg_max // 2
    if deploy:
        b1, b2 = b1.item(), b2.item()
        step = (b1 + 1) ** (2 / (reg_max - 2))
        vals = [-b2] + [-(step ** i) + 1 for i in range(half - 1, 0, -1)] + [0.0]
        vals += [step ** i - 1 for i in range(1, half)] + [b2]
        return torch.tensor(vals, dtype=up.dtype, device=up.device)
    step = (b1 + 1) ** (2 / (reg_max - 2))
    neg = [-(step ** i) + 1 for i in range(half - 1, 0, -1)]
    pos = [step ** i - 1 for i in range(1, half)]
    return torch.cat([-b2] + neg + [torch.zeros_like(up[0][None])] + pos + [b2], 0)


def distance2bbox(points, distance, reg_scale):
    """Edge distances (in units of w/reg_scale resp. h/reg_scale, offset by 0.5*reg_scale)
    around a cxcywh reference box -> cxcywh box  (ref arch/utils.py:119-142)."""
    rs = abs(reg_scale)
    sx, sy = points[..., 2] / rs, points[..., 3] / rs
    x1 = points[..., 0] - (0.5 * rs + distance[..., 0]) * sx
    y1 = points[..., 1] - (0.5 * rs + distance[..., 1]) * sy
    x2 = points[..., 0] + (0.5 * rs + distance[..., 2]) * sx
    y2 = points[..., 1] + (0.5 * rs + distance[..., 3]) * sy
    return box_xyxy_to_cxcywh(torch.stack([x1, y1, x2, y2], -1))


def translate_gt(gt, reg_max, reg_scale, up):
    """Continuous distances -> (left bin index as float, right weight, left weight)
    (ref arch/utils.py:267-325)."""
    gt = gt.reshape(-1)
    w = weighting_function(reg_max, up, reg_scale)
    left = ((w[None, :] - gt[:, None]) <= 0).sum(1) - 1  # last bin with W <= gt
    idx = left.float()
    inside = (idx >= 0) & (idx < reg_max)
    li = left.clamp(0, reg_max - 1)
    dl = (gt - w[li]).abs()
    dr = (w[li + 1] - gt).abs()
    wr_in = dl / (dl + dr)
    below, above = idx < 0, idx >= reg_max
    zero, one = torch.zeros_like(idx), torch.ones_like(idx)
    weight_right = torch.where(inside, wr_in, torch.where(above, one, zero))
    weight_left = torch.where(inside, 1.0 - wr_in, torch.where(below, one, zero))
    idx = torch.where(below, zero, idx)
    idx = torch.where(above, torch.full_like(idx, reg_max - 0.1), idx)
    return idx, weight_right, weight_left


def bbox2distance(points, bbox, reg_max, reg_scale, up, eps=0.1):
    """FGL targets: xyxy GT around cxcywh reference points -> bin index / interpolation
    weights per edge  (ref arch/utils.py:328-354)."""
    rs = abs(reg_scale)
    ux = points[..., 2] / rs + 1e-16
    uy = points[..., 3] / rs + 1e-16
    left = (points[:, 0] - bbox[:, 0]) / ux - 0.5 * rs
    top = (points[:, 1] - bbox[:, 1]) / uy - 0.5 * rs
    right = (bbox[:, 2] - points[:, 0]) / ux - 0.5 * rs
    bottom = (bbox[:, 3] - points[:, 1]) / uy - 0.5 * rs
    d, wr, wl = translate_gt(torch.stack([left, top, right, bottom], -1), reg_max, reg_scale, up)
    if reg_max is not None:
        d = d.clamp(min=0, max=reg_max - eps)
    return d.reshape(-1).detach(), wr.detach(), wl.detach()


# ------------------------------------------------------------------ deformable attention
def deformable_attention_core_func_v2(
    value, value_spatial_shapes, sampling_locations: Tensor, attention_weights: Tensor,
    num_points_list: List[int], method="default",
):
    """Drop-in for the reference's function of the same name (arch/utils.py:191-264).

    value: list of per-level tensors [bs, n_head, c, h*w] (the reference's layout) OR a single
    tensor [bs, sum(h*w), n_head, c] (this build's native, permute-free layout).
    sampling_locations [bs, Lq, n_head, sum(points), 2] in [0,1]; attention_weights
    [bs, Lq, n_head, sum(points)].  Returns [bs, Lq, n_head*c].
    """
    if method != "default":
        raise NotImplementedError("only the bilinear ('default') sampling method is built")
    if isinstance(value, (list, tuple)):
        value = torch.cat(list(value), dim=-1).permute(0, 3, 1, 2).contiguous()
    return kernels.msda(value, value_spatial_shapes, sampling_locations, attention_weights,
                        num_points_list)


# ------------------------------------------------------------------ contrastive denoising
_NOISE_GENERATOR = None


def set_denoising_generator(gen):
    """Parity hook: draw the denoising noise from the CPU generator `gen` (then move it to the
    targets' device) instead of the device RNG, so that a CPU run and a GPU run - or the
    reference and this build - see identical noise.  `None` restores the device RNG."""
    global _NOISE_GENERATOR
    _NOISE_GENERATOR = gen


CAPTURE_KEEP = None      # a hip.CaptureArena while dl.engine.GraphedSegment captures: pinned staging of the uploads recorded in the graph


def upload(array, device):
    """Host numpy array / CPU tensor -> device without stalling the host: pinned staging + async copy for
    CUDA devices (a pageable-memory copy waits for the device queue to drain), plain conversion on CPU.
    Inside a HIP-graph capture the copy becomes a memcpy node that re-reads the host buffer at every replay: it is staged
    in the capturing segment's own pinned arena (CAPTURE_KEEP), alive and unchanged for the graph's lifetime, and copied
    by the library (torch's pinned copies record an allocator event on the capturing stream)."""
    t = torch.from_numpy(array) if isinstance(array, np.ndarray) else array
    if torch.device(device).type != "cuda":
        return t.to(device)
    if CAPTURE_KEEP is not None:
        return CAPTURE_KEEP.upload(t, device)
    return t.pin_memory().to(device, non_blocking=True)


def _rand_like(t, dtype=None):
    if _NOISE_GENERATOR is None:
        return torch.rand_like(t, dtype=dtype)
    return torch.rand(t.shape, generator=_NOISE_GENERATOR, dtype=dtype or t.dtype).to(t.device)


def _randint_like(t, low, high, dtype=None):
    if _NOISE_GENERATOR is None:
        return torch.randint_like(t, low, high, dtype=dtype)
    return torch.randint(low, high, t.shape, generator=_NOISE_GENERATOR,
                         dtype=dtype or t.dtype).to(t.device)


_CDN_MASKS = {}


def _cdn_attn_mask(total, num_queries, g, device):
    """Boolean self-attention mask of the denoising + matching queries (True = may not attend; ref arch/utils.py:442-455): a
    function of (total, num_queries, group size) only, so it is built once per shape and SHARED between steps - callers must not
    write to it (the attention kernels' bit-packed / tile-summary forms are cached on the tensor's address and version too)."""
    key = (int(total), int(num_queries), int(g), str(device))
    mask = _CDN_MASKS.get(key)
    if mask is None:
        if len(_CDN_MASKS) >= 64:
            _CDN_MASKS.clear()
        n = total + num_queries
        mask = torch.zeros([n, n], dtype=torch.bool, device=device)
        mask[total:, :total] = True  # matching queries never see the reconstruction part
        gid = torch.arange(total, device=device) // g
        mask[:total, :total] = gid[:, None] != gid[None, :]  # groups are mutually invisible
        _CDN_MASKS[key] = mask
    return mask


def _cdn_group_torch(targets, counts, device, bs, gmax, groups, num_classes, label_noise_ratio, box_noise_scale):
    """The padded class ids with label noise and the noised boxes in logit space as torch ops (CPU tensors, off-default noise
    settings; the GPU runs csrc/cdn.hip, which restates exactly this sequence)."""
    # pad labels / boxes to [bs, gmax].  The slot of every real GT is known from the host-side counts, so
    # the padded tensors are filled with an index_copy through ONE async upload - a boolean-mask assignment
    # or torch.tensor(list, device=cuda) would each block the host until the device queue drains.
    slots = np.concatenate([i * gmax + np.arange(n, dtype=np.int64) for i, n in enumerate(counts)])
    slots = upload(slots, device)
    valid = torch.zeros(bs * gmax, dtype=torch.bool, device=device)
    cls = torch.full([bs * gmax], num_classes, dtype=torch.int32, device=device)
    box = torch.zeros([bs * gmax, 4], device=device)
    if sum(counts):
        valid.index_fill_(0, slots, True)
        cls.index_copy_(0, slots, torch.cat([t["labels"] for t in targets]).to(torch.int32))
        box.index_copy_(0, slots, torch.cat([t["boxes"] for t in targets]).to(box.dtype))
    valid, cls, box = valid.view(bs, gmax), cls.view(bs, gmax), box.view(bs, gmax, 4)

    cls = cls.tile([1, 2 * groups])
    box = box.tile([1, 2 * groups, 1])
    valid = valid.tile([1, 2 * groups])
    neg = torch.zeros([bs, gmax * 2, 1], device=device)
    neg[:, gmax:] = 1
    neg = neg.tile([1, groups, 1])
    if label_noise_ratio > 0:
        flip = _rand_like(cls, dtype=torch.float) < (label_noise_ratio * 0.5)
        rnd = _randint_like(flip, 0, num_classes, dtype=cls.dtype)
        cls = torch.where(flip & valid, rnd, cls)

    if box_noise_scale > 0:
        xyxy = box_cxcywh_to_xyxy(box)
        span = torch.tile(box[..., 2:] * 0.5, [1, 1, 2]) * box_noise_scale
        sign = _randint_like(box, 0, 2) * 2.0 - 1.0
        mag = _rand_like(box)
        mag = (mag + 1.0) * neg + mag * (1 - neg)
        xyxy = torch.clip(xyxy + sign * mag * span, min=0.0, max=1.0)
        box = box_xyxy_to_cxcywh(xyxy)
        box = torch.where(box < 0, -box, box)
    box_unact = inverse_sigmoid(box)
    return cls, box_unact


def get_contrastive_denoising_training_group(
    targets, num_classes, num_queries, class_embed, num_denoising=100,
    label_noise_ratio=0.5, box_noise_scale=1.0,
):
    """Builds the CDN query group (ref arch/utils.py:357-467).

    Same RNG draw order / shapes / dtypes as the reference (rand_like(int32->float),
    randint_like(bool->int32), randint_like(box), rand_like(box)), so a fixed CPU seed
    reproduces the reference's noise; the per-image python loops are replaced by padded
    tensor ops.
    """
    if num_denoising <= 0:
        return None, None, None, None
    counts = [len(t["labels"]) for t in targets]
    device = targets[0]["labels"].device
    gmax = max(counts)
    if gmax == 0:
        return None, None, None, {"dn_positive_idx": None, "dn_num_group": 0,
                                  "dn_num_split": [0, num_queries]}
    groups = max(num_denoising // gmax, 1)
    bs = len(counts)

    # positive (first-half) slots of every group that hold a real GT; known from the counts alone,
    # so built on the host (the reference derives them with a device nonzero + split)
    pos_np = [(np.arange(groups, dtype=np.int64)[:, None] * (2 * gmax) + np.arange(n, dtype=np.int64)[None, :]).reshape(-1)
              for n in counts]
    pos_idx = tuple(torch.from_numpy(p) for p in pos_np)
    total = int(gmax * 2 * groups)

    if device.type == "cuda" and label_noise_ratio > 0 and box_noise_scale > 0 and kernels.cdn_kernel_enabled():
        # GPU: the padded labels / boxes, the label flips and the box noise in ONE launch (csrc/cdn.hip, bit-identical to the
        # composition below); the four random tensors are drawn here, in the reference's order, shapes and dtypes
        offsets = upload(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device)
        labels_cat = torch.cat([t["labels"] for t in targets]).to(torch.int64)
        boxes_cat = torch.cat([t["boxes"] for t in targets]).to(torch.float32)
        like_cls = torch.empty([bs, total], dtype=torch.int32, device=device)
        like_box = torch.empty([bs, total, 4], dtype=torch.float32, device=device)
        flip_rand = _rand_like(like_cls, dtype=torch.float)
        rnd = _randint_like(like_cls.view(torch.int32), 0, num_classes, dtype=torch.int32)
        sign01 = _randint_like(like_box, 0, 2)
        mag = _rand_like(like_box)
        cls, box_unact = kernels.cdn_group(labels_cat, boxes_cat, offsets, flip_rand, rnd, sign01, mag, bs, gmax, groups,
                                           num_classes, label_noise_ratio * 0.5, box_noise_scale)
    else:
        cls, box_unact = _cdn_group_torch(targets, counts, device, bs, gmax, groups, num_classes, label_noise_ratio,
                                          box_noise_scale)

    logits = kernels.embedding(class_embed, cls) if isinstance(class_embed, nn.Embedding) else class_embed(cls)

    mask = _cdn_attn_mask(total, num_queries, gmax * 2, device)

    meta = {"dn_positive_idx": pos_idx, "dn_num_group": groups, "dn_num_split": [total, num_queries],
            "dn_positive_flat": np.concatenate(pos_np)}
    return logits, box_unact, mask, meta
