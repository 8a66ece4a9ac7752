"""Differentiable fp32 torch (CPU) versions of the HIP-backed operators - TEST ORACLE ONLY.

`install()` plugs them into `custom_d_fine_amd.kernels` so that the host logic (model /
criterion) can run on CPU tensors: used by the CPU test-suite, by __graft_entry__.smoke() as
the checker, and by bench.py's cpu_baseline leg.  The product never calls install().
Each operator is written from the reference's maths independently of the HIP kernels (and of
F.grid_sample, which the reference uses), and is itself checked against oracle/np_ref.py and
the golden vectors generated from the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ref


def _level_table(shapes, points, device):
    start, hh, ww = np_ref._level_table(shapes, points)
    return (torch.as_tensor(start, device=device), torch.as_tensor(hh, device=device),
            torch.as_tensor(ww, device=device))


def msda(value, shapes, loc, weight, points):
    """Explicit 4-corner bilinear gather (reference: arch/utils.py:191-264)."""
    B, L, H, D = value.shape
    Lq, P = loc.shape[1], loc.shape[3]
    start, hh, ww = _level_table(shapes, points, value.device)
    v32 = value.float()
    x = loc[..., 0].float() * ww - 0.5
    y = loc[..., 1].float() * hh - 0.5
    x0, y0 = x.floor(), y.floor()
    fx, fy = x - x0, y - y0
    flat = v32.permute(0, 2, 1, 3).reshape(B * H, L, D)               # [(b h), L, D]
    out = 0
    for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1)):
        xi, yi = (x0 + dx).long(), (y0 + dy).long()
        wgt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
        ok = (xi >= 0) & (xi < ww) & (yi >= 0) & (yi < hh)
        rows = start + yi.clamp(min=0).minimum(hh - 1) * ww + xi.clamp(min=0).minimum(ww - 1)
        rows = rows.permute(0, 2, 1, 3).reshape(B * H, Lq * P)
        g = flat.gather(1, rows[..., None].expand(-1, -1, D)).reshape(B, H, Lq, P, D)
        coef = (wgt * ok * weight.float()).permute(0, 2, 1, 3)         # [B,H,Lq,P]
        out = out + (coef[..., None] * g).sum(3)
    return out.permute(0, 2, 1, 3).reshape(B, Lq, H * D).to(value.dtype)


def msda_fused(value, shapes, ref, offsets, logits, points, offset_scale):
    """dfine_decoder.py:147,156-166 followed by msda()."""
    scale = torch.tensor([1.0 / n for n in points for _ in range(n)], device=value.device)
    ref = ref.float()[:, :, None, None, :]
    loc = ref[..., :2] + offsets.float() * scale[None, None, None, :, None] * ref[..., 2:] * offset_scale
    return msda(value, shapes, loc, F.softmax(logits.float(), -1), points)


def hungarian_assign(logits, boxes, tgt_labels, tgt_boxes, sizes, w_class, w_bbox, w_giou,
                     alpha, gamma, use_focal=True, extra_cost=None):
    """Per (head, image): numpy cost block (np_ref.match_cost) + C LSAP (oracle/lsap.c)."""
    assert use_focal, "oracle covers the focal-cost matcher only (reference default)"
    K, B, Q, _ = logits.shape
    lg, bx = logits.detach().cpu().numpy(), boxes.detach().cpu().numpy()
    ids, tb = tgt_labels.cpu().numpy(), tgt_boxes.detach().cpu().numpy()
    T = int(sum(sizes))
    tmax = max(sizes) if sizes else 0
    cols = np.full((K, T), -1, np.int32)
    cost_out = np.zeros((K, B, Q, tmax), np.float32)
    for k in range(K):
        off = 0
        for b, n in enumerate(sizes):
            if n:
                c = np_ref.match_cost(lg[k, b], bx[k, b], ids[off:off + n], tb[off:off + n],
                                      w_class=w_class, w_bbox=w_bbox, w_giou=w_giou,
                                      alpha=alpha, gamma=gamma)
                if extra_cost is not None:
                    c = c + extra_cost[k, b, :, :n].detach().cpu().numpy()
                    c = np.nan_to_num(c, nan=1.0).astype(np.float32)
                cost_out[k, b, :, :n] = c
                r, t = np_ref.lsap(c)
                cols[k, off + t] = r
            off += n
    return torch.from_numpy(cols), torch.from_numpy(cost_out)


def detection_topk(logits, boxes, k, height, width, to_round=True):
    """DFINEPostProcessor's focal branch restated (reference src/dl/export.py:35-59 box arithmetic, :61-84 top-k):
    sigmoid -> topk over Q*C -> idx % C, idx // C -> gathered absolute xyxy boxes."""
    B, Q, C = logits.shape
    b = boxes.float().reshape(-1, 4)
    xc, yc, bw, bh = b[:, 0] * width, b[:, 1] * height, b[:, 2] * width, b[:, 3] * height
    x0, y0, x1, y1 = xc - bw / 2, yc - bh / 2, xc + bw / 2, yc + bh / 2
    if to_round:
        x0, y0 = torch.clamp(torch.floor(x0), min=1), torch.clamp(torch.floor(y0), min=1)
        x1, y1 = torch.clamp(torch.ceil(x1), max=width - 1), torch.clamp(torch.ceil(y1), max=height - 1)
    else:
        x0, y0 = torch.clamp(x0, min=0), torch.clamp(y0, min=0)
        x1, y1 = torch.clamp(x1, max=width), torch.clamp(y1, max=height)
    abs_boxes = torch.stack([x0, y0, x1, y1], 1).view(B, Q, 4)
    scores, idx = torch.topk(torch.sigmoid(logits.float()).flatten(1), k, dim=-1)
    labels, qidx = idx % C, idx // C
    return labels, qidx, abs_boxes.gather(1, qidx.unsqueeze(-1).expand(-1, -1, 4)), scores


def preprocess_frames(frames, out_hw, resized_hw, top_left=(0, 0), pad_value=114, dtype=torch.float32):
    """Torch_model._preprocess restated on the host (reference src/infer/torch_model.py:240-298,378-418): resize with
    np_ref.resize_linear_u8, constant border, BGR->RGB, HWC->CHW, /255."""
    f = frames.cpu().numpy()
    out = np.full((f.shape[0], out_hw[0], out_hw[1], 3), pad_value, dtype=np.uint8)
    for i in range(f.shape[0]):
        r = f[i] if tuple(resized_hw) == f[i].shape[:2] else np_ref.resize_linear_u8(f[i], resized_hw[0], resized_hw[1])
        out[i, top_left[0]: top_left[0] + resized_hw[0], top_left[1]: top_left[1] + resized_hw[1]] = r
    t = torch.from_numpy(out[..., ::-1].transpose(0, 3, 1, 2).copy())
    return (t.float() / 255.0).to(dtype)


def install():
    """Route CPU tensors of the HIP-backed operators to this module (tests / cpu baseline)."""
    import sys
    from custom_d_fine_amd import kernels
    kernels._TEST_BACKEND = sys.modules[__name__]


def uninstall():
    from custom_d_fine_amd import kernels
    kernels._TEST_BACKEND = None
