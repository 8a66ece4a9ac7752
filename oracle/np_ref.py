"""numpy restatement of the reference arithmetic behind every HIP kernel - TEST ORACLE ONLY.

Nothing here is imported by `custom_d_fine_amd`.  All functions are float32 unless noted and
cite the reference (`/root/reference/src/d_fine/...`) lines they restate.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_lsap.restype = ctypes.c_int
    return _LIB


# ---------------------------------------------------------------------------------- A12 LSAP
def lsap(cost):
    """scipy.optimize.linear_sum_assignment restated in C (oracle/lsap.c); reference call sites
    matcher.py:243,264.  cost [nr, nc] any float dtype -> (rows i64 ascending, cols i64)."""
    c = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = c.shape
    n = min(nr, nc)
    rows = np.zeros(n, np.int64)
    cols = np.zeros(n, np.int64)
    if n == 0:
        return rows, cols
    ret = _lib().oracle_lsap(ctypes.c_int64(nr), ctypes.c_int64(nc),
                             c.ctypes.data_as(ctypes.c_void_p),
                             rows.ctypes.data_as(ctypes.c_void_p),
                             cols.ctypes.data_as(ctypes.c_void_p))
    if ret == -2:
        raise ValueError("matrix contains invalid numeric entries")
    if ret == -1:
        raise ValueError("cost matrix is infeasible")
    return rows, cols


# ---------------------------------------------------------------------------------- A7 MSDA
def _level_table(shapes, points):
    """per sampling point: (level start row in value, H, W)."""
    starts, hs, ws = [], [], []
    off = 0
    for (h, w), n in zip(shapes, points):
        starts += [off] * n
        hs += [h] * n
        ws += [w] * n
        off += h * w
    return np.array(starts), np.array(hs), np.array(ws)


def msda_forward(value, shapes, loc, weight, points):
    """deformable_attention_core_func_v2 (arch/utils.py:191-264) for method="default":
    per level F.grid_sample(value_l, 2*loc-1, bilinear, zeros, align_corners=False), i.e.
    pixel coordinate = loc*size - 0.5, then sum_p weight * sample.

    value [B, L, H, D] (row l of level k = start_k + y*W_k + x); loc [B, Lq, H, P, 2] (x, y);
    weight [B, Lq, H, P] -> out [B, Lq, H*D] float32."""
    value = np.asarray(value, np.float32)
    loc = np.asarray(loc, np.float32)
    weight = np.asarray(weight, np.float32)
    B, L, H, D = value.shape
    _, Lq, _, P, _ = loc.shape
    start, hh, ww = _level_table(shapes, points)
    x = loc[..., 0] * ww.astype(np.float32) - np.float32(0.5)
    y = loc[..., 1] * hh.astype(np.float32) - np.float32(0.5)
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    out = np.zeros((B, Lq, H, D), np.float32)
    bidx = np.arange(B)[:, None, None, None]
    hidx = np.arange(H)[None, None, :, None]
    for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1)):
        xi = (x0 + dx).astype(np.int64)
        yi = (y0 + dy).astype(np.int64)
        wgt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
        ok = (xi >= 0) & (xi < ww) & (yi >= 0) & (yi < hh)
        rows = start + np.clip(yi, 0, hh - 1) * ww + np.clip(xi, 0, ww - 1)
        v = value[bidx, rows, hidx]                      # [B, Lq, H, P, D]
        out += ((wgt * ok * weight)[..., None] * v).sum(3)
    return out.reshape(B, Lq, H * D)


def msda_backward(value, shapes, loc, weight, points, grad_out):
    """Analytic gradients of msda_forward wrt value, loc, weight (what autograd derives from
    grid_sample_backward + mul/sum in the reference).  Returns (g_value, g_loc, g_weight)."""
    value = np.asarray(value, np.float32)
    loc = np.asarray(loc, np.float32)
    weight = np.asarray(weight, np.float32)
    B, L, H, D = value.shape
    _, Lq, _, P, _ = loc.shape
    go = np.asarray(grad_out, np.float32).reshape(B, Lq, H, 1, D)
    start, hh, ww = _level_table(shapes, points)
    x = loc[..., 0] * ww.astype(np.float32) - np.float32(0.5)
    y = loc[..., 1] * hh.astype(np.float32) - np.float32(0.5)
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    g_value = np.zeros_like(value)
    g_w = np.zeros_like(weight)
    g_x = np.zeros_like(weight)
    g_y = np.zeros_like(weight)
    bidx = np.broadcast_to(np.arange(B)[:, None, None, None], weight.shape)
    hidx = np.broadcast_to(np.arange(H)[None, None, :, None], weight.shape)
    for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1)):
        xi = (x0 + dx).astype(np.int64)
        yi = (y0 + dy).astype(np.int64)
        wx = fx if dx else 1 - fx
        wy = fy if dy else 1 - fy
        ok = ((xi >= 0) & (xi < ww) & (yi >= 0) & (yi < hh)).astype(np.float32)
        rows = start + np.clip(yi, 0, hh - 1) * ww + np.clip(xi, 0, ww - 1)
        v = value[bidx, rows, hidx]                      # [B, Lq, H, P, D]
        dot = (v * go).sum(-1) * ok                      # <grad_out, corner value>
        g_w += wx * wy * dot
        g_x += (1.0 if dx else -1.0) * wy * dot * weight
        g_y += (1.0 if dy else -1.0) * wx * dot * weight
        contrib = (wx * wy * ok * weight)[..., None] * go
        np.add.at(g_value, (bidx, rows, hidx), contrib)
    g_loc = np.stack([g_x * ww.astype(np.float32), g_y * hh.astype(np.float32)], -1)
    return g_value, g_loc.astype(np.float32), g_w


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def msda_prologue(ref, offsets, logits, points, offset_scale=0.5):
    """MSDeformableAttention.forward, 4-d reference-box branch (dfine_decoder.py:147,156-166):
    loc = ref_xy + offsets * (1/n_level) * ref_wh * offset_scale; w = softmax over points."""
    scale = np.array([1.0 / n for n in points for _ in range(n)], np.float32)
    ref = np.asarray(ref, np.float32)[:, :, None, None, :]            # [B,Lq,1,1,4]
    off = np.asarray(offsets, np.float32) * scale[None, None, None, :, None] * ref[..., 2:] * \
        np.float32(offset_scale)
    return (ref[..., :2] + off).astype(np.float32), softmax(np.asarray(logits, np.float32), -1)


# ---------------------------------------------------------------------------------- A11 costs
def box_cxcywh_to_xyxy(b):
    b = np.asarray(b, np.float32)
    w = np.maximum(b[..., 2], 0) * np.float32(0.5)
    h = np.maximum(b[..., 3], 0) * np.float32(0.5)
    return np.stack([b[..., 0] - w, b[..., 1] - h, b[..., 0] + w, b[..., 1] + h], -1)


def pairwise_iou_giou(a, b):
    """box_iou / generalized_box_iou (arch/utils.py:12-51) on xyxy boxes a [N,4], b [M,4]."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    iou = inter / union
    lt2 = np.minimum(a[:, None, :2], b[None, :, :2])
    rb2 = np.maximum(a[:, None, 2:], b[None, :, 2:])
    wh2 = np.clip(rb2 - lt2, 0, None)
    hull = wh2[..., 0] * wh2[..., 1]
    return iou, iou - (hull - union) / hull


def match_cost(logits, boxes, tgt_ids, tgt_boxes, w_class=2.0, w_bbox=5.0, w_giou=2.0,
               alpha=0.25, gamma=2.0):
    """HungarianMatcher cost block of ONE image (matcher.py:135-169,242), focal variant:
    C = w_bbox * cdist_1(box, tgt) + w_class * (pos - neg) + w_giou * (-GIoU); NaN -> 1.
    logits [Q, C], boxes [Q, 4] cxcywh, tgt_ids [T], tgt_boxes [T, 4] -> [Q, T] float32."""
    x = np.asarray(logits, np.float32)
    p = (np.float32(1) / (np.float32(1) + np.exp(-x)))[:, np.asarray(tgt_ids)]
    a, g = np.float32(alpha), np.float32(gamma)
    neg = (1 - a) * (p ** g) * (-np.log(1 - p + np.float32(1e-8)))
    pos = a * ((1 - p) ** g) * (-np.log(p + np.float32(1e-8)))
    c_class = pos - neg
    bx = np.asarray(boxes, np.float32)
    tb = np.asarray(tgt_boxes, np.float32)
    c_bbox = np.abs(bx[:, None, :] - tb[None, :, :]).sum(-1)
    _, giou = pairwise_iou_giou(box_cxcywh_to_xyxy(bx), box_cxcywh_to_xyxy(tb))
    c = np.float32(w_bbox) * c_bbox + np.float32(w_class) * c_class + np.float32(w_giou) * (-giou)
    c = c.astype(np.float32)
    c[np.isnan(c)] = 1.0
    c[np.isposinf(c)] = np.finfo(np.float32).max
    c[np.isneginf(c)] = np.finfo(np.float32).min
    return c


def hungarian(logits, boxes, tgt_ids, tgt_boxes, **kw):
    """One image of HungarianMatcher.forward: cost block -> LSAP -> (query idx, target idx)."""
    if len(tgt_ids) == 0:
        z = np.zeros(0, np.int64)
        return z, z
    return lsap(match_cost(logits, boxes, tgt_ids, tgt_boxes, **kw))


# ---------------------------------------------------------------------------------- A8 FDR
def weighting_function(reg_max, up, reg_scale):
    """W(n) (arch/utils.py:145-188), float32 like the reference's tensor arithmetic."""
    up = np.float32(abs(up))
    rs = np.float32(abs(reg_scale))
    b1 = up * rs
    b2 = b1 * np.float32(2)
    step = np.float32((b1 + np.float32(1)) ** np.float32(2 / (reg_max - 2)))
    half = reg_max // 2
    neg = [-(step ** np.float32(i)) + np.float32(1) for i in range(half - 1, 0, -1)]
    pos = [step ** np.float32(i) - np.float32(1) for i in range(1, half)]
    return np.array([-b2] + neg + [0.0] + pos + [b2], np.float32)


def integral(corners, project):
    """Integral.forward (dfine_decoder.py:291-295): softmax over reg_max+1 bins . W(n)."""
    nb = project.shape[0]
    p = softmax(np.asarray(corners, np.float32).reshape(-1, nb), -1)
    return (p @ project.astype(np.float32)).reshape(list(corners.shape[:-1]) + [-1])


def distance2bbox(points, distance, reg_scale):
    """arch/utils.py:119-142."""
    rs = np.float32(abs(reg_scale))
    p = np.asarray(points, np.float32)
    d = np.asarray(distance, np.float32)
    x1 = p[..., 0] - (np.float32(0.5) * rs + d[..., 0]) * (p[..., 2] / rs)
    y1 = p[..., 1] - (np.float32(0.5) * rs + d[..., 1]) * (p[..., 3] / rs)
    x2 = p[..., 0] + (np.float32(0.5) * rs + d[..., 2]) * (p[..., 2] / rs)
    y2 = p[..., 1] + (np.float32(0.5) * rs + d[..., 3]) * (p[..., 3] / rs)
    return np.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


def resize_linear_u8(img, out_h, out_w):
    """OpenCV's 8-bit bilinear resize restated (cv2.resize(..., INTER_LINEAR) on uint8, imgproc/resize.cpp of opencv-python -
    a pip dependency of the reference, requirements.txt, not vendored in /root/reference and absent here: PARITY UNPINNED
    against cv2 itself; call site: src/infer/torch_model.py:246-248,406).  img uint8 [H, W, C] -> uint8 [out_h, out_w, C].
    Pixel centres aligned, 11-bit fixed-point coefficients, horizontal pass in int32, vertical pass
    (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2."""
    h, w = img.shape[:2]

    def coef(n_out, n_in):
        d = np.arange(n_out, dtype=np.float64)
        f = ((d + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= n_in - 1
        f[hi], s[hi] = 0.0, n_in - 1
        a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return s, np.minimum(s + 1, n_in - 1), a0, a1

    x0, x1, ax0, ax1 = coef(out_w, w)
    y0, y1, ay0, ay1 = coef(out_h, h)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]            # [H, out_w, C]
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


# ----------------------------------------------------------------------------------------------------------------------
# (f3) augmentation geometry - restated from the reference's host code (src/dl/utils.py), which cannot be imported here
# (it imports cv2 / albumentations at module level).
def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.1, eps=1e-16):
    """Reference utils.py:283-295: box1 / box2 [4, n] before / after the augmentation."""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def affine_boxes(boxes, M, scale, target_size, area_thr=0.1):
    """Reference utils.py:343-377 (random_affine without polygons): boxes [n, 4] xyxy through the 3 x 3 matrix M by their four
    corners, min / max, clip to [0, target_w] x [0, target_h] -> (new boxes [n, 4] f32, keep [n] bool)."""
    n = len(boxes)
    xy = np.ones((n * 4, 3), dtype=np.float32)
    xy[:, :2] = boxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = (xy @ np.asarray(M, dtype=np.float64).T)[:, :2].reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    new = np.stack([x.min(1), y.min(1), x.max(1), y.max(1)], axis=1)
    new[:, [0, 2]] = new[:, [0, 2]].clip(0, target_size[0])
    new[:, [1, 3]] = new[:, [1, 3]].clip(0, target_size[1])
    keep = box_candidates(box1=boxes.T * scale, box2=new.T, area_thr=area_thr)
    return new.astype(np.float32), keep


def warp_affine_u8(src, M2x3, out_hw, border=114):
    """cv2.warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT) restated (OpenCV imgwarp.cpp: inverse map
    in AB_BITS = 10 fixed point, INTER_BITS = 5 sub-pixel positions, bilinear weights from the 2^15-scaled table whose rows are
    normalised by adjusting their largest entry).  cv2 is not in the build image: parity with cv2 itself is UNPINNED; this
    restatement is what the HIP kernel is checked against, next to size-independent properties."""
    Hs, Ws = src.shape[:2]
    Hd, Wd = out_hw
    m = np.asarray(M2x3, dtype=np.float64).reshape(2, 3)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    i00, i01, i10, i11 = m[1, 1] * D, -m[0, 1] * D, -m[1, 0] * D, m[0, 0] * D
    b1, b2 = -i00 * m[0, 2] - i01 * m[1, 2], -i10 * m[0, 2] - i11 * m[1, 2]
    xs, ys = np.arange(Wd), np.arange(Hd)
    adelta, bdelta = np.rint(i00 * xs * 1024).astype(np.int64), np.rint(i10 * xs * 1024).astype(np.int64)
    X0 = np.rint((i01 * ys + b1) * 1024).astype(np.int64) + 16
    Y0 = np.rint((i11 * ys + b2) * 1024).astype(np.int64) + 16
    X, Y = (X0[:, None] + adelta[None, :]) >> 5, (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    wx1, wy1 = fx.astype(np.float32) / 32, fy.astype(np.float32) / 32
    w = np.stack([(1 - wy1) * (1 - wx1), (1 - wy1) * wx1, wy1 * (1 - wx1), wy1 * wx1], -1).astype(np.float32)
    wi = np.rint(w * np.float32(32768)).astype(np.int64)
    diff = 32768 - wi.sum(-1)
    kmax = wi.argmax(-1)
    np.put_along_axis(wi, kmax[..., None], np.take_along_axis(wi, kmax[..., None], -1) + diff[..., None], -1)
    out = np.zeros((Hd, Wd, 3), dtype=np.int64)
    for k, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        px, py = sx + dx, sy + dy
        ok = (px >= 0) & (px < Ws) & (py >= 0) & (py < Hs)
        v = np.where(ok[..., None], src[np.clip(py, 0, Hs - 1), np.clip(px, 0, Ws - 1)].astype(np.int64), border)
        out += v * wi[..., k][..., None]
    return np.clip((out + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


# =============================================================================================
# torch.argsort(counts, descending=True) on the CPU - the tie order of the GO-index vote (ref dfine_criterion.py:570-591 calls
# it per image on the multiplicities of the (query, target) pairs).  ATen's CPU sort (aten/src/ATen/native/cpu/SortingKernel.cpp,
# torch==2.9.0 pinned by requirements.txt:33; same code in the container's 2.10.0) runs std::sort over a composite
# (value, index) iterator with the comparator "lhs.value > rhs.value": libstdc++'s introsort - median-of-three quicksort down
# to runs of 16, heapsort past 2 * floor(log2 n) levels, one final insertion sort.  Unstable, but deterministic: restated here
# step by step (bits/stl_algo.h, bits/stl_heap.h) so that the device kernel (csrc/plans.hip) can be pinned against it and it
# against torch itself (tests/test_plans.py).
# =============================================================================================
def aten_argsort_desc(counts):
    v = [int(c) for c in counts]
    ix = list(range(len(v)))

    def comp(a, b):                     # KeyValueCompDesc on positions
        return v[a] > v[b]

    def swap(a, b):
        v[a], v[b] = v[b], v[a]
        ix[a], ix[b] = ix[b], ix[a]

    def unguarded_linear_insert(last):
        val, vi = v[last], ix[last]
        nxt = last - 1
        while val > v[nxt]:
            v[last], ix[last] = v[nxt], ix[nxt]
            last = nxt
            nxt -= 1
        v[last], ix[last] = val, vi

    def insertion_sort(first, last):
        if first == last:
            return
        for i in range(first + 1, last):
            if comp(i, first):
                val, vi = v[i], ix[i]
                v[first + 1:i + 1] = v[first:i]
                ix[first + 1:i + 1] = ix[first:i]
                v[first], ix[first] = val, vi
            else:
                unguarded_linear_insert(i)

    def push_heap(first, hole, top, val, vi):
        parent = (hole - 1) // 2
        while hole > top and v[first + parent] > val:
            v[first + hole], ix[first + hole] = v[first + parent], ix[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        v[first + hole], ix[first + hole] = val, vi

    def adjust_heap(first, hole, length, val, vi):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if comp(first + child, first + child - 1):
                child -= 1
            v[first + hole], ix[first + hole] = v[first + child], ix[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[first + hole], ix[first + hole] = v[first + child - 1], ix[first + child - 1]
            hole = child - 1
        push_heap(first, hole, top, val, vi)

    def heap_sort(first, last):         # std::__partial_sort(first, last, last): make_heap + sort_heap
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, v[first + parent], ix[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            val, vi = v[last], ix[last]
            v[last], ix[last] = v[first], ix[first]
            adjust_heap(first, 0, last - first, val, vi)

    def move_median_to_first(result, a, b, c):
        if comp(a, b):
            if comp(b, c):
                swap(result, b)
            elif comp(a, c):
                swap(result, c)
            else:
                swap(result, a)
        elif comp(a, c):
            swap(result, a)
        elif comp(b, c):
            swap(result, c)
        else:
            swap(result, b)

    def unguarded_partition(first, last, pivot):
        while True:
            while comp(first, pivot):
                first += 1
            last -= 1
            while comp(pivot, last):
                last -= 1
            if not first < last:
                return first
            swap(first, last)
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    n = len(v)
    if n:
        introsort_loop(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion_sort(0, 16)
            for i in range(16, n):
                unguarded_linear_insert(i)
        else:
            insertion_sort(0, n)
    return np.asarray(ix, dtype=np.int64)
