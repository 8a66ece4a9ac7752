"""Torch-facing operators of the hot path.

Every function here takes/returns torch tensors and, for CUDA(ROCm) tensors, launches the
hand-written HIP kernels of `csrc/` through the C ABI (`custom_d_fine_amd.hip`, ctypes) on
torch's current stream.  There is NO CPU implementation in the product: for a CPU tensor the
HIP-backed operators raise, unless a test harness has installed the oracle backend
(`oracle.torch_backend.install()` - used only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg).

Operators that are still composed from ATen calls (the fp32 math of configs[1], the mask head of configs[4])
are marked "ATen plumbing" - they run the same code on CPU and GPU and are the next ones to be
replaced by HIP kernels.
"""
import os
import weakref
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

_TEST_BACKEND = None  # set by oracle.torch_backend.install(); never by product code


def _backend_for_cpu(op: str):
    if _TEST_BACKEND is None:
        raise RuntimeError(
            f"custom_d_fine_amd.kernels.{op}: got a CPU tensor. This operator only exists as a "
            "HIP kernel for gfx950 (MI355X); move the tensors to the GPU. (CPU execution is "
            "available to the test-suite only, through oracle.torch_backend.install().)")
    return getattr(_TEST_BACKEND, op)


_HIP = None


def _library_fallback(what):
    """A CUDA tensor under bf16 autocast reached an operator form the build has no kernel for.  The product does not drop to a
    library (MIOpen / hipBLASLt / the SDPA kernels) silently: it raises, unless DFINE_ALLOW_LIBRARY=1 asks for the ATen
    composition (comparison runs, shapes outside the reference's configurations)."""
    if _env("DFINE_ALLOW_LIBRARY", "0") != "1":
        raise RuntimeError(f"custom_d_fine_amd: no HIP kernel for {what} (bf16 autocast on the GPU); "
                           "set DFINE_ALLOW_LIBRARY=1 to run the ATen / library composition instead")


def _bf16_autocast():
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16


def _hip():
    global _HIP
    if _HIP is None:                       # (a plain global: the import statement costs ~1 us per call, ~350 calls per forward pass)
        from . import hip  # raises loudly if libdfine_hip.so is missing / not loadable
        _HIP = hip
    return _HIP


# =============================================================================================
# A7  multi-scale deformable attention gather
# =============================================================================================
class _MSDA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, loc, weight, shapes, points):
        hip = _hip()
        value = value.contiguous()
        loc = loc.float().contiguous()
        weight = weight.float().contiguous()
        out = hip.msda_forward(value, loc, weight, shapes, points)
        ctx.save_for_backward(value, loc, weight)
        ctx.shapes, ctx.points = shapes, points
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, loc, weight = ctx.saved_tensors
        gv, gl, gw = _hip().msda_backward(value, loc, weight, grad_out.contiguous(),
                                          ctx.shapes, ctx.points)
        return gv, gl, gw, None, None


def msda(value: torch.Tensor, spatial_shapes, sampling_locations: torch.Tensor,
         attention_weights: torch.Tensor, num_points_list: List[int]) -> torch.Tensor:
    """value [B, sum(HW), H, hd]; sampling_locations [B, Lq, H, P, 2] in [0,1] (x, y);
    attention_weights [B, Lq, H, P]; P = sum(num_points_list).  -> [B, Lq, H*hd] (value dtype).
    Bilinear, zero padding, align_corners=False (pixel centre at (i+0.5)/size)."""
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    points = tuple(int(p) for p in num_points_list)
    if not value.is_cuda:
        return _backend_for_cpu("msda")(value, shapes, sampling_locations, attention_weights, points)
    return _MSDA.apply(value, sampling_locations, attention_weights, shapes, points)


class _ValueGradShare:
    """All decoder layers gather from the same `value` (the encoder memory, split by heads once: ref
    dfine_decoder.py:410-420,443): their backward kernels add d(value) into ONE fp32 accumulator that is zero-filled once
    and rounded to the value dtype once, instead of a zero fill + cast per layer and three adds of the results by autograd.
    Every forward use registers here; the backward call that brings the count back to zero returns the total, the others
    return None.  A backward pass that skips some of the registered uses would lose their share, so the count is
    per-graph: `begin()` (called where `value` is made) starts a fresh one."""

    def __init__(self):
        self.pending = 0
        self.acc = None


_OWNED_VALUE_GRAD = [None]      # address of the value-path gradient _MSDAFused's shared accumulator has just produced


def msda_share_value_grad(value):
    """Marks `value` as gathered by several msda_fused calls of one forward pass (the decoder calls it once per step)."""
    if value.is_cuda and value.requires_grad:
        value._dfine_share = _ValueGradShare()
    return value


class _MSDAFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, ref, offsets, logits, shapes, points, offset_scale):
        hip = _hip()
        share = getattr(value, "_dfine_share", None)
        value = value.contiguous()
        ref = ref.float().contiguous()
        offsets = offsets.contiguous()
        logits = logits.contiguous()
        out = hip.msda_fused_forward(value, ref, offsets, logits, shapes, points, offset_scale)
        ctx.save_for_backward(value, ref, offsets, logits)
        ctx.cfg = (shapes, points, offset_scale)
        ctx.share = share if (share is not None and ctx.needs_input_grad[0]) else None
        if ctx.share is not None:
            ctx.share.pending += 1
        return out

    @staticmethod
    def backward(ctx, grad_out):
        hip = _hip()
        value, ref, offsets, logits = ctx.saved_tensors
        shapes, points, offset_scale = ctx.cfg
        share = ctx.share
        if share is not None and share.pending <= 0:      # a second backward through the same graph: plain path
            share = None
        if share is None:
            gv, goff, glog = hip.msda_fused_backward(
                value, ref, offsets, logits, grad_out.contiguous(), shapes, points, offset_scale)
            return gv, None, goff, glog, None, None, None
        if share.acc is None:
            share.acc = hip.msda_grad_value_buffer(value, uses=share.pending)
        _, goff, glog = hip.msda_fused_backward(value, ref, offsets, logits, grad_out.contiguous(), shapes, points,
                                                offset_scale, gv_acc=share.acc)
        share.pending -= 1
        gv = None
        if share.pending == 0:
            gv, share.acc = hip.msda_finish_grad_value(share.acc, value.dtype), None
            _OWNED_VALUE_GRAD[0] = gv.data_ptr()         # a fresh tensor of this pass (see _TakeRowsAndPass.backward)
        return gv, None, goff, glog, None, None, None


def msda_fused(value, spatial_shapes, ref_boxes, offsets, logits, num_points_list,
               offset_scale: float = 0.5):
    """Deformable gather with the location arithmetic and the point-softmax fused in:
         loc = ref_xy + offsets * (1/points_of_level) * ref_wh * offset_scale
         w   = softmax_P(logits)
    value [B, L, H, hd]; ref_boxes [B, Lq, 4] cxcywh (no gradient: the decoder detaches them);
    offsets [B, Lq, H, P, 2]; logits [B, Lq, H, P].  -> [B, Lq, H*hd]."""
    shapes = tuple((int(h), int(w)) for h, w in spatial_shapes)
    points = tuple(int(p) for p in num_points_list)
    if not value.is_cuda:
        return _backend_for_cpu("msda_fused")(value, shapes, ref_boxes, offsets, logits, points,
                                              float(offset_scale))
    return _MSDAFused.apply(value, ref_boxes.detach(), offsets, logits, shapes, points,
                            float(offset_scale))


# =============================================================================================
# A3  encoder maps -> decoder token memory
# =============================================================================================
class _EmbeddingSideGrad(torch.autograd.Function):
    """nn.Embedding lookup whose weight gradient runs on the side stream next to the decoder's backward chain (hip._side_fork;
    joined by the fused optimizer's gather like the other gradient tensors produced there) and, for small tables, as ONE scan
    kernel (dfine_embedding_bwd) instead of ATen's sort-based embedding_dense_backward (radix sort + segmented scatter: 0.17-0.37 ms
    for the denoising class embedding of a D-FINE-m step)."""

    @staticmethod
    def forward(ctx, weight, idx, padding_idx):
        ctx.save_for_backward(idx)
        ctx.cfg = (weight.shape[0], padding_idx, weight)
        return F.embedding(idx, weight, padding_idx)

    @staticmethod
    def backward(ctx, g):
        hip = _hip()
        (idx,) = ctx.saved_tensors
        n, padding_idx, weight = ctx.cfg
        ctx.cfg = None
        pad = -1 if padding_idx is None else padding_idx
        g = g.contiguous()
        small = n <= 1024 and g.dtype == torch.float32 and idx.dtype in (torch.int32, torch.int64)
        if _side_wgrad_ok(weight) and hip.side_stream_ok():
            st = hip._side_fork(g.device)
            with torch.cuda.stream(st.stream):
                dw = (hip.embedding_backward(g, idx, n, pad, stream=st.cuda_stream) if small
                      else torch.ops.aten.embedding_dense_backward(g, idx, n, pad, False))
            hip._SIDE_LIVE.append((g, idx))
            return dw, None, None
        if small:
            return hip.embedding_backward(g, idx, n, pad), None, None
        return torch.ops.aten.embedding_dense_backward(g, idx, n, pad, False), None, None


def embedding(mod: nn.Embedding, idx):
    """mod(idx) for a plain nn.Embedding; on the GPU training path the weight gradient leaves the backward chain."""
    if (idx.is_cuda and mod.weight.requires_grad and torch.is_grad_enabled() and mod.max_norm is None and not mod.sparse
            and not mod.scale_grad_by_freq and _env("DFINE_HIP_UNITS", "1") == "1"):
        return _EmbeddingSideGrad.apply(mod.weight, idx, mod.padding_idx)
    return mod(idx)


HEAD_ARENA = True      # the criterion's head-loss output blocks as slices of one zero-filled arena (tools/ab_step.py kernels.HEAD_ARENA)
CDN_KERNEL = True      # (tools/ab_step.py kernels.CDN_KERNEL)
STACK_LAYER_OUTPUTS = False      # the decoder hands its per-layer head outputs on as lists, not as stacked tensors (see TransformerDecoder.forward)


def cdn_kernel_enabled():
    return CDN_KERNEL and _env("DFINE_HIP_UNITS", "1") == "1"


def cdn_group(labels, boxes, offsets, flip_rand, rnd_cls, sign01, mag, bs, gmax, groups, num_classes, flip_below, box_noise_scale):
    """Padded class ids with label noise + noised boxes in logit space of the denoising group, one launch (csrc/cdn.hip); no
    gradient flows through it (the reference builds it from the targets)."""
    with torch.no_grad():
        return _hip().cdn_group(labels, boxes, offsets, flip_rand, rnd_cls, sign01, mag, bs, gmax, groups, num_classes,
                                flip_below, box_noise_scale)


class _Upsample2Nearest(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode="nearest") of a bf16 NCHW map (FPN top-down path): one data-movement pass each way
    (csrc/layout.hip) instead of ATen's gather kernels (78 / 73 us for the 40x40 -> 80x80 map of D-FINE-m bs 32)."""

    @staticmethod
    def forward(ctx, x):
        return _hip().upsample2_nearest(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return _hip().upsample2_nearest(dy.contiguous(), backward=True)


def upsample2_nearest(x):
    if x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] % 4 == 0 and _env("DFINE_HIP_UNITS", "1") == "1":
        return _Upsample2Nearest.apply(x)
    return F.interpolate(x, scale_factor=2.0, mode="nearest")


class _FlattenLevels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bufs, *maps):
        maps = tuple(m.contiguous() for m in maps)
        ctx.shapes = [(m.shape[2], m.shape[3]) for m in maps]
        ctx.bufs = bufs
        return _hip().maps_to_tokens(maps)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        return (None,) + tuple(_hip().tokens_to_maps(g, ctx.shapes, outs=ctx.bufs))


def flatten_levels(maps):
    """concat([m.flatten(2).permute(0, 2, 1) for m in maps], 1): the decoder's memory [B, L, C] from the encoder's maps
    (ref dfine_decoder.py:778-801).  GPU/bf16: tiled transposes (csrc/layout.hip); the backward hands every level a
    contiguous gradient map instead of a strided view of d(memory)."""
    if (maps[0].is_cuda and all(m.dtype == torch.bfloat16 and m.dim() == 4 for m in maps)
            and maps[0].shape[1] % 8 == 0 and all((m.shape[2] * m.shape[3]) % 8 == 0 and m.shape[1] == maps[0].shape[1] for m in maps)
            and _env("DFINE_HIP_UNITS", "1") == "1"):
        # a map that came out of a captured segment (dl.engine.GraphedSegment) carries the static buffer its gradient is
        # expected in: the backward pass writes there directly instead of into a fresh map that is copied afterwards
        bufs = tuple(getattr(m, "_dfine_grad_buf", None) for m in maps)
        return _FlattenLevels.apply(bufs if any(b is not None for b in bufs) else None, *maps)
    return torch.concat([m.flatten(2).permute(0, 2, 1) for m in maps], 1)


# =============================================================================================
# A11/A12  matcher: cost matrix + linear sum assignment, all heads of a step in one launch
# =============================================================================================
def hungarian_assign(logits: torch.Tensor, boxes: torch.Tensor, tgt_labels: torch.Tensor,
                     tgt_boxes: torch.Tensor, sizes: List[int], w_class: float, w_bbox: float,
                     w_giou: float, alpha: float, gamma: float, use_focal: bool = True,
                     extra_cost: Optional[torch.Tensor] = None):
    """logits [K, B, Q, C], boxes [K, B, Q, 4] (K prediction heads matched against the same
    targets); tgt_labels [T] i64, tgt_boxes [T, 4] concatenated over the batch, `sizes` = targets
    per image.  Returns (cols, cost): `cols` int32 [K, T] = the query assigned to every target
    (-1 when an image has more targets than queries), `cost` [K, B, Q, Tmax] or None."""
    if not logits.is_cuda:
        return _backend_for_cpu("hungarian_assign")(
            logits, boxes, tgt_labels, tgt_boxes, sizes, w_class, w_bbox, w_giou, alpha, gamma,
            use_focal, extra_cost)
    return _hip().hungarian_assign(logits, boxes, tgt_labels, tgt_boxes, sizes, w_class, w_bbox,
                                   w_giou, alpha, gamma, use_focal, extra_cost)


# =============================================================================================
# A8  FDR head: Integral + distance2bbox + LQE statistics
# =============================================================================================
class _FDRDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, corners, ref, wtable, reg_scale):
        hip = _hip()
        lead = corners.shape[:-1]
        c = corners.contiguous()
        r = ref.detach().float().contiguous()
        boxes, stat, idx = hip.fdr_forward(c, r, wtable, reg_scale)
        ctx.save_for_backward(c, r, idx)
        ctx.cfg = (wtable, reg_scale)
        ctx.mark_non_differentiable(idx)
        return boxes.view(*lead, 4), stat.view(*lead, stat.shape[-1])

    @staticmethod
    def backward(ctx, g_boxes, g_stat):
        c, r, idx = ctx.saved_tensors
        wtable, reg_scale = ctx.cfg
        gb = None if g_boxes is None else g_boxes.float().contiguous()
        gs = None if g_stat is None else g_stat.float().contiguous()
        return _hip().fdr_backward(c, r, wtable, reg_scale, gb, gs, idx), None, None, None


def fdr_decode(corners, ref_boxes, wtable, reg_scale):
    """corners [..., 4*(reg_max+1)] edge-distribution logits, ref_boxes [..., 4] (detached) ->
    (boxes [..., 4] f32 cxcywh, stat [..., 4*(k+1)] f32: top-4 bin probabilities + mean per edge).
    GPU only (csrc/fdr.hip); CPU tensors use the Integral / distance2bbox / LQE torch composition."""
    if not corners.is_cuda:
        raise RuntimeError("kernels.fdr_decode is a HIP operator")
    return _FDRDecode.apply(corners, ref_boxes, wtable, reg_scale)


# =============================================================================================
# A13/A14  set-criterion losses of one prediction head (values + gradients in one launch group)
# =============================================================================================
HEAD_GRADS_FUSED = True       # backward of the fused head losses as one in-place launch (tools/ab_step.py kernels.HEAD_GRADS_FUSED)


class _HeadLosses(torch.autograd.Function):
    """-> tensor[5] = (vfl, l1, giou, fgl, ddf), already weighted and normalised.  The kernels
    compute the closed-form gradients in the forward pass; backward only scales them."""

    @staticmethod
    def forward(ctx, logits, boxes, corners, ref, teacher_corners, teacher_logits, cls_plan,
                box_plan, tgt_labels, tgt_boxes, cfg):
        hip = _hip()
        if boxes.dtype != torch.float32:
            boxes = boxes.float()
        dt = logits.dtype

        def same(t):
            return None if t is None else (t if t.dtype == dt else t.to(dt))

        corners_k, tc, tl = same(corners), same(teacher_corners), same(teacher_logits)
        out, g_logits, g_l1, g_giou, g_fgl, g_ddf = hip.head_losses(
            logits, boxes, corners_k, None if ref is None else ref.float(), tc, tl, cls_plan,
            box_plan, tgt_labels, tgt_boxes, cfg["wtable"], cfg["reg_max"], cfg["reg_scale"],
            cfg["alpha"], cfg["gamma"], cfg["temp"], cfg["s_vfl"], cfg["s_l1"], cfg["s_giou"],
            cfg["s_fgl"], cfg["c_pos"], cfg["c_neg"], scales_dev=cfg.get("scales_dev"), box_count_dev=cfg.get("box_count_dev"),
            zbytes=cfg.get("zbytes"))
        ctx.grads = (g_logits, g_l1, g_giou, g_fgl, g_ddf)
        ctx.dtypes = (boxes.dtype, None if corners is None else corners.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        g_logits, g_l1, g_giou, g_fgl, g_ddf = ctx.grads
        ctx.grads = None
        if (HEAD_GRADS_FUSED and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and ctx.dtypes[0] == torch.float32
                and (g_fgl is None or ctx.dtypes[1] == g_fgl.dtype)):
            # one in-place launch instead of ~8 element-wise ones per head (the saved gradients are used once)
            _hip().head_grads_scale(g, g_logits, g_l1, g_giou, g_fgl, g_ddf)
            return g_logits, g_l1, g_fgl, None, None, None, None, None, None, None, None
        d_logits = g_logits * g[0].to(g_logits.dtype)
        d_boxes = g_l1 * g[1] + g_giou * g[2]
        d_corners = None
        if g_fgl is not None:
            d_corners = g_fgl * g[3].to(g_fgl.dtype)
            if g_ddf is not None:
                d_corners = d_corners + g_ddf * g[4].to(g_ddf.dtype)
            if ctx.dtypes[1] is not None and d_corners.dtype != ctx.dtypes[1]:
                d_corners = d_corners.to(ctx.dtypes[1])
        return d_logits, d_boxes, d_corners, None, None, None, None, None, None, None, None


def head_losses(logits, boxes, corners, ref, teacher_corners, teacher_logits, cls_plan, box_plan,
                tgt_labels, tgt_boxes, cfg):
    """Fused VFL + L1/GIoU + FGL + DDF of one head on the GPU (csrc/losses.hip).  logits [B,Q,C],
    boxes [B,Q,4], corners/teacher_* [B,Q,4*(reg_max+1)] or None may be strided views;
    *_plan int64 [3, M] (image, query, target row).  No CPU form: the criterion composes the same
    terms from torch ops for CPU tensors."""
    if not logits.is_cuda:
        raise RuntimeError("kernels.head_losses is a HIP operator (CPU tensors use the torch composition "
                           "in DFINECriterion)")
    return _HeadLosses.apply(logits, boxes, corners, ref, teacher_corners, teacher_logits, cls_plan,
                             box_plan, tgt_labels, tgt_boxes, cfg)


# =============================================================================================
# A1/A2  conv -> BatchNorm -> activation -> learnable affine units of backbone and encoder
# =============================================================================================
def _stem_wgrad_side(weight):
    # (measured equal, 32.47 / 32.57 vs 32.58 / 32.51 ms per step: the weight gradients of the layers whose backward runs LAST -
    # stem1, the 2x2 kernels - kept on the main stream, which has nothing left to do by then)
    return _side_wgrad_ok(weight)


def _side_wgrad_ok(weight):
    """May this parameter's gradient TENSOR be produced on the side stream (hip._side_fork)?  Only when its sole consumer is the
    fused optimizer's gather (which joins the side stream first): the parameter is managed by it, fp32 (no cast kernel on
    the main stream) and has no gradient yet (autograd then just stores the tensor instead of adding to it)."""
    slot = getattr(weight, "_dfine_slot", None)
    return slot is not None and slot[0].defer_wgrads and weight.dtype == torch.float32 and weight.grad is None


class _DepthwiseConv(torch.autograd.Function):
    """Depthwise k x k conv, NCHW (HIP: dwconv.hip).  Weights stay fp32 master parameters."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, fanin=None):
        """fanin: GradFanIn of x - this layer is the consumer that runs its backward LAST and adds onto the parked gradient."""
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad)
        ctx.fanin = None
        if fanin is not None and ctx.needs_input_grad[0] and _hip().dwconv_acc_supported(x, weight.shape[-1], stride, pad):
            fanin.armed = True
            ctx.fanin = fanin
        return _hip().dwconv_forward(x, weight.detach().float().contiguous(), stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        fan, ctx.fanin = ctx.fanin, None
        acc = fan.take() if (fan is not None and fan.parking) else None
        dx, dw = _hip().dwconv_backward(x, weight.detach().float().contiguous(), dy, stride, pad,
                                        ctx.needs_input_grad[0], ctx.needs_input_grad[1], side_dw=_side_wgrad_ok(weight), acc=acc)
        return dx, (dw.to(weight.dtype) if dw is not None else None), None, None, None


class _BNAct(torch.autograd.Function):
    """BatchNorm2d (+ReLU/SiLU) (+scalar affine) in one op (HIP: bnact.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, lab_scale, lab_bias, running_mean, running_var, act, training,
                momentum, eps):
        x = x.contiguous()
        y, stats = _hip().bn_act_forward(x, gamma, beta, running_mean, running_var, lab_scale, lab_bias,
                                         act, training, momentum, eps)
        ctx.save_for_backward(x, stats, lab_scale)
        # (parameter gradients only where a parameter wants one: the frozen units of the D-FINE-l / x backbone carry buffers, and an
        # eval-mode backward without them skips the reduction pass over x and dy altogether)
        ctx.cfg = (act, training, gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]),
                   lab_scale is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]))
        # learnable affine: the backward kernel adds its two scalars straight into the fused optimizer's flat gradient
        # buffer when scale and bias sit next to each other there (they do: consecutive parameters of one module)
        ctx.slot = None
        if lab_scale is not None and lab_bias is not None and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]:
            slot = _defer_slot(lab_scale, lab_bias)
            if slot is not None and slot[0].grad_offset(slot[1][1]) == slot[0].grad_offset(slot[1][0]) + 1:
                ctx.slot = slot
                for i in slot[1]:
                    slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, lab_scale = ctx.saved_tensors
        act, training, has_affine, has_lab = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        slot = ctx.slot
        dlab_ptr = slot[0].grad_ptr(slot[1][0]) if slot is not None else None
        dx, dg, db, dlab = _hip().bn_act_backward(x, dy, stats, lab_scale, act, training, has_affine, has_lab, dlab_ptr)
        dls = dlb = None
        if slot is not None:
            for i in slot[1]:
                slot[0].use_done(i)
        elif has_lab:
            dls, dlb = dlab[0:1], dlab[1:2]
        return dx, dg, db, dls, dlb, None, None, None, None, None, None


# Per-shape choice between the HIP implicit-GEMM kernels and MIOpen for forward / data gradient /
# weight gradient of a dense conv: measured once per (shape, op) on first use ("measure, don't guess":
# the HIP kernels win on every 3x3 and on the small latency-bound 1x1 layers, MIOpen's GEMM-like
# kernels win on the large 1x1 layers - profiles/r01_conv_survey_hip_vs_miopen.txt).
_CONV_PLAN = {}

# Environment switches are read once (an os.environ lookup per layer call is ~1 us x 800 calls per step);
# `reload_env()` re-reads them (tests / tools that flip a switch after import).
_ENV = {}


def _env(name, default):
    v = _ENV.get(name)
    if v is None:
        v = _ENV[name] = os.environ.get(name, default)
    return v


def reload_env():
    global _ROUTES
    _ENV.clear()
    _ROUTES = [0]
    _CONV_PLAN.clear()

# Packed bf16 copies of the conv weights (forward and data-gradient layouts) are rebuilt only when the
# weights changed: `_WEIGHT_EPOCH` is bumped by the optimizer step (the fused optimizer updates the flat
# buffer through raw pointers, invisible to tensor._version), `_version` covers in-place torch updates.
_WEIGHT_EPOCH = 0
_PACK_CACHE = {}


def bump_weight_epoch():
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


_CAPTURE_POSSIBLE = False      # set by dl.engine.GraphedSegment: only then is the capture query worth a call per layer
_CAPTURE_FROZEN_WEIGHTS = False   # set by infer.Torch_model while it captures: the weights never change again, so the cached
                                  # packed / bf16 copies (filled by its eager warm-up) are served inside the capture
_CAPTURE_SHADOWS = False          # set by dl.engine.GraphedSegment while it captures a TRAINING segment: the registered packed /
                                  # bf16 copies live at fixed addresses and the segment refreshes all of them (one launch per kind,
                                  # `refresh_weight_shadows`) in front of every replay, so they are served inside the capture too


def _recording_packs():
    """True inside a HIP-graph capture whose replays see changed weights that nobody refreshes outside the graph: the pack /
    cast launch itself must then be recorded, and the caches neither served nor filled."""
    return (_CAPTURE_POSSIBLE and not _CAPTURE_FROZEN_WEIGHTS and not _CAPTURE_SHADOWS
            and torch.cuda.is_current_stream_capturing())


def _capturing_training_segment():
    return _CAPTURE_SHADOWS and torch.cuda.is_current_stream_capturing()


def refresh_weight_shadows(device):
    """Brings every registered packed / bf16 / transposed-bf16 weight copy up to date with the master weights (at most one
    launch per kind, nothing when no optimizer step happened since the last refresh).  The eager path does this lazily on
    the first request after a step; a captured segment (dl.engine.GraphedSegment) calls it in front of every replay."""
    if _PACK_REGISTRY and _PACK_BATCH_EPOCH != _WEIGHT_EPOCH:
        _pack_all_registered(device)
    if _BF16_REGISTRY and _BF16_EPOCH != _WEIGHT_EPOCH:
        _refresh_bf16_params(device)
    if _BF16T_REGISTRY and _BF16T_EPOCH != _WEIGHT_EPOCH:
        _refresh_bf16_t(device)


# Every (weight, layout) pair the model has asked for is remembered; the first request after an optimizer step
# repacks ALL of them with one launch (their master weights are views of the flat parameter buffer and the
# packed buffers persist, so the device table is built once) instead of one pack launch per layer and layout.
_PACK_REGISTRY = {}        # key -> [weakref(weight), dgrad, packed tensor, data_ptr, version]
_PACK_TABLE = None         # (device table, n_entries, registry size when built)
_PACK_BATCH_EPOCH = -1


def _pack_all_registered(device):
    global _PACK_TABLE, _PACK_BATCH_EPOCH
    import numpy as np
    hip = _hip()
    dead = [k for k, e in _PACK_REGISTRY.items() if e[0]() is None]
    for k in dead:
        del _PACK_REGISTRY[k]
        _PACK_TABLE = None
    stale = _PACK_TABLE is None or _PACK_TABLE[2] != len(_PACK_REGISTRY)
    if not stale:
        for e in _PACK_REGISTRY.values():
            if e[0]().data_ptr() != e[3]:
                stale = True
                break
    if stale:
        rows = []
        for e in _PACK_REGISTRY.values():
            w = e[0]()
            cout, cin, ks, _ = w.shape
            n, kk = (cin, cout) if e[1] else (cout, cin)
            rows.append((w.data_ptr(), e[2].data_ptr(), cout, cin, ks, (n + 15) // 16 * 16, (kk + 31) // 32 * 32, int(e[1])))
            e[3] = w.data_ptr()
        from .d_fine.arch.utils import upload
        _PACK_TABLE = (upload(np.asarray(rows, dtype=np.int64), device), len(rows), len(_PACK_REGISTRY))
    hip.conv_pack_weights_multi(_PACK_TABLE[0], _PACK_TABLE[1])
    for e in _PACK_REGISTRY.values():
        e[4] = e[0]()._version
    _PACK_BATCH_EPOCH = _WEIGHT_EPOCH


def _packed_weights(weight, dgrad):
    if _recording_packs():
        # inside a HIP-graph capture the pack launch itself must be recorded (the weights change
        # between replays), so never serve or fill the cache here
        return _hip().conv_pack_weights(weight.detach().float().contiguous(), dgrad)
    key = (id(weight), dgrad)
    ent = _PACK_REGISTRY.get(key)
    batchable = weight.dtype == torch.float32 and weight.is_contiguous()
    if ent is not None and ent[0]() is weight and batchable:
        if _PACK_BATCH_EPOCH != _WEIGHT_EPOCH:
            if _capturing_training_segment():
                raise RuntimeError("packed weights are stale inside a segment capture (refresh_weight_shadows was not called)")
            _pack_all_registered(weight.device)
        if ent[4] == weight._version and ent[3] == weight.data_ptr():
            return ent[2]
        if _capturing_training_segment():
            return _hip().conv_pack_weights(weight.detach().float().contiguous(), dgrad)     # recorded, cache untouched
        # updated in place by torch since the batch pack (plain optimizers): repack this one
        ent[2].copy_(_hip().conv_pack_weights(weight.detach(), dgrad))
        ent[3], ent[4] = weight.data_ptr(), weight._version
        return ent[2]
    w2 = _hip().conv_pack_weights(weight.detach().float().contiguous(), dgrad)
    if batchable and not _capturing_training_segment():       # (a copy made inside a capture lives in the graph's pool)
        global _PACK_TABLE
        # a new or REPLACED entry (id() reuse after an earlier model was freed) invalidates the device pointer table:
        # neither len() nor the data_ptr check of _pack_all_registered would notice a replaced key
        _PACK_REGISTRY[key] = [weakref.ref(weight), dgrad, w2, weight.data_ptr(), weight._version]
        _PACK_TABLE = None
    return w2


# bf16 shadow copies of fp32 parameters (what autocast would re-cast at every use): all registered parameters are
# refreshed with ONE launch on the first request after an optimizer step (same protocol as the packed conv weights).
_BF16_REGISTRY = {}        # id(param) -> [weakref(param), bf16 copy, data_ptr, version]
_BF16_TABLE = None
_BF16_EPOCH = -1


def _refresh_bf16_params(device):
    global _BF16_TABLE, _BF16_EPOCH
    import numpy as np
    dead = [k for k, e in _BF16_REGISTRY.items() if e[0]() is None]
    for k in dead:
        del _BF16_REGISTRY[k]
        _BF16_TABLE = None
    stale = _BF16_TABLE is None or _BF16_TABLE[2] != len(_BF16_REGISTRY)
    if not stale:
        for e in _BF16_REGISTRY.values():
            if e[0]().data_ptr() != e[2]:
                stale = True
                break
    if stale:
        rows = []
        for e in _BF16_REGISTRY.values():
            p = e[0]()
            rows.append((p.data_ptr(), e[1].data_ptr(), p.numel()))
            e[2] = p.data_ptr()
        from .d_fine.arch.utils import upload
        _BF16_TABLE = (upload(np.asarray(rows, dtype=np.int64), device), len(rows), len(_BF16_REGISTRY))
    _hip().multi_cast_bf16(_BF16_TABLE[0], _BF16_TABLE[1])
    for e in _BF16_REGISTRY.values():
        e[3] = e[0]()._version
    _BF16_EPOCH = _WEIGHT_EPOCH


def bf16_param(p):
    """bf16 copy of an fp32 CUDA parameter, cached until the weights change."""
    if p.dtype == torch.bfloat16:
        return p.detach()
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) or _recording_packs():
        return p.detach().to(torch.bfloat16)
    ent = _BF16_REGISTRY.get(id(p))
    if ent is not None and ent[0]() is p:
        seg = _capturing_training_segment()
        if _BF16_EPOCH != _WEIGHT_EPOCH:
            if seg:
                raise RuntimeError("bf16 weight shadows are stale inside a segment capture (refresh_weight_shadows was not called)")
            _refresh_bf16_params(p.device)
        if ent[3] != p._version or ent[2] != p.data_ptr():       # changed in place by plain torch code
            if seg:
                return p.detach().to(torch.bfloat16)
            ent[1].copy_(p.detach())
            ent[2], ent[3] = p.data_ptr(), p._version
        return ent[1]
    if _capturing_training_segment():
        return p.detach().to(torch.bfloat16)                    # recorded in the graph, cache untouched
    c = p.detach().to(torch.bfloat16)
    global _BF16_TABLE
    _BF16_REGISTRY[id(p)] = [weakref.ref(p), c, p.data_ptr(), p._version]
    _BF16_TABLE = None                 # inserted or replaced entry: rebuild the device pointer table
    return c


# Transposed bf16 shadows ([K, N] of an nn.Linear weight [N, K]): the B^T operand of the data-gradient GEMM
# dX = dY . W (csrc/gemm.hip).  Same refresh protocol: one launch for all registered weights after an optimizer step.
_BF16T_REGISTRY = {}       # id(param) -> [weakref(param), bf16 [K, N] copy, data_ptr, version]
_BF16T_TABLE = None
_BF16T_EPOCH = -1


def _refresh_bf16_t(device):
    global _BF16T_TABLE, _BF16T_EPOCH
    import numpy as np
    dead = [k for k, e in _BF16T_REGISTRY.items() if e[0]() is None]
    for k in dead:
        del _BF16T_REGISTRY[k]
        _BF16T_TABLE = None
    stale = _BF16T_TABLE is None
    if not stale:
        for e in _BF16T_REGISTRY.values():
            if e[0]().data_ptr() != e[2]:
                stale = True
                break
    if stale:
        rows = []
        for e in _BF16T_REGISTRY.values():
            p = e[0]()
            rows.append((p.data_ptr(), e[1].data_ptr(), p.shape[0], p.shape[1]))
            e[2] = p.data_ptr()
        from .d_fine.arch.utils import upload
        _BF16T_TABLE = (upload(np.asarray(rows, dtype=np.int64), device), len(rows))
    _hip().multi_cast_bf16_t(_BF16T_TABLE[0], _BF16T_TABLE[1])
    for e in _BF16T_REGISTRY.values():
        e[3] = e[0]()._version
    _BF16T_EPOCH = _WEIGHT_EPOCH


def bf16_param_t(p):
    """bf16 TRANSPOSE [K, N] of a 2-d fp32 CUDA parameter [N, K], cached until the weights change."""
    global _BF16T_TABLE
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 2) or _recording_packs():
        return p.detach().t().to(torch.bfloat16).contiguous()
    ent = _BF16T_REGISTRY.get(id(p))
    if ent is not None and ent[0]() is p:
        seg = _capturing_training_segment()
        if _BF16T_EPOCH != _WEIGHT_EPOCH:
            if seg:
                raise RuntimeError("bf16 weight shadows are stale inside a segment capture (refresh_weight_shadows was not called)")
            _refresh_bf16_t(p.device)
        if ent[3] != p._version or ent[2] != p.data_ptr():       # changed in place by plain torch code
            if seg:
                return p.detach().t().to(torch.bfloat16).contiguous()
            ent[1].copy_(p.detach().t())
            ent[2], ent[3] = p.data_ptr(), p._version
        return ent[1]
    if _capturing_training_segment():
        return p.detach().t().to(torch.bfloat16).contiguous()    # recorded in the graph, cache untouched
    c = p.detach().t().to(torch.bfloat16).contiguous()
    _BF16T_REGISTRY[id(p)] = [weakref.ref(p), c, p.data_ptr(), p._version]
    _BF16T_TABLE = None
    return c


def _conv_plan(x, weight):
    """{'fwd','dgrad','wgrad'}: the HIP kernels serve every dense 1x1 / 3x3 stride-1 convolution `_mfma_conv_ok` lets through
    (weight gradients of narrow planes on zero-padded copies, hip.conv_wgrad_bf16).  The per-shape timing against MIOpen of
    rounds 1-3 lives in tools/conv_survey.py; the product has one route."""
    H, W = x.shape[2], x.shape[3]
    ks = weight.shape[-1]
    key = (H, W, ks)
    plan = _CONV_PLAN.get(key)
    if plan is None:
        ok = _hip().conv_wgrad_supported(H, W, ks)
        if not ok:
            raise RuntimeError(f"custom_d_fine_amd: no HIP weight-gradient kernel for a {ks}x{ks} convolution on {H}x{W} maps")
        plan = _CONV_PLAN[key] = {"fwd": True, "dgrad": True, "wgrad": True}
    return plan


_FUSE_CONV_BN = True      # conv node + BatchNorm tail as ONE autograd node (tests may flip it to compare the two-node composition)


def _conv_plan_all_hip(x, weight):
    _conv_plan(x, weight)
    return True


def _defer_slot(*params):
    """(fused optimizer, [indices]) when every given parameter lives in a FusedAdamWEMA flat buffer that takes deferred weight
    gradients (split partial sums reduced by ONE launch per step straight into the flat gradient buffer), else None."""
    fused, idx = None, []
    for p in params:
        slot = getattr(p, "_dfine_slot", None)
        if slot is None or not slot[0].defer_wgrads or (fused is not None and slot[0] is not fused) or p.dtype != torch.float32:
            return None
        fused = slot[0]
        idx.append(slot[1])
    return fused, idx


class _DenseConv(torch.autograd.Function):
    """1x1 / 3x3 stride-1 dense convolution, NCHW bf16: forward / data gradient / weight gradient on the HIP implicit-GEMM
    MFMA kernels (conv.hip)."""

    @staticmethod
    def forward(ctx, x, weight):
        hip = _hip()
        x = x.contiguous()
        _conv_plan(x, weight)
        ks = weight.shape[-1]
        y = hip.conv_forward_bf16(x, _packed_weights(weight, False), weight.shape[0], ks)
        ctx.save_for_backward(x, weight)
        ctx.slot = _defer_slot(weight) if ctx.needs_input_grad[1] else None
        if ctx.slot is not None:
            ctx.slot[0].note_use(ctx.slot[1][0])
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        ks = weight.shape[-1]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_dx:
            w2 = (hip.conv_pack_weights(weight.detach().float().contiguous(), True) if getattr(ctx, "direct", False)
                  else _packed_weights(weight, True))
            dx = hip.conv_forward_bf16(dy, w2, weight.shape[1], ks)
        if need_dw:
            slot = getattr(ctx, "slot", None)
            if slot is not None:
                ws, meta = hip.conv_wgrad_bf16(x, dy, ks, partials=True)
                slot[0].defer_wgrad(slot[1][0], ws, meta)
                slot[0].use_done(slot[1][0])
            else:
                dw = hip.conv_wgrad_bf16(x, dy, ks).to(weight.dtype)
        return dx, dw


class GradFanIn:
    """Hand-off of a data gradient between the two consumers of one map inside a block (HG_Block: layer i output ->
    layer i + 1 and the aggregation conv, ref hgnetv2.py:265-274).  The reference lets autograd add the two gradients (one
    element-wise pass, 2 reads + 1 write of the map).  Here the consumer that runs its backward FIRST (the aggregation: every
    layer's gradient depends on it) parks its gradient in `buf` and reports None to autograd; the consumer that runs LATER adds
    its own data gradient onto the parked one in its convolution's epilogue (dfine_conv_accum_bf16) and returns the sum.
    `armed` is set in the forward pass by the later consumer when it will be able to do that, `parking` by the earlier one
    (whose forward runs after it) when it will park - only then does the later consumer expect a parked gradient."""
    __slots__ = ("armed", "parking", "buf", "chain", "lo", "pending")

    def __init__(self, chain=False, lo=0):
        self.armed, self.parking, self.buf = False, False, None
        # chain role (fan_slice below): SEVERAL later consumers, all of them 1x1 convolutions reading channels lo.. of the map;
        # each adds its data gradient onto that channel range of the parked gradient and leaves it parked; `pending` counts them
        self.chain, self.lo, self.pending = chain, lo, 0

    def take(self):
        buf, self.buf = self.buf, None
        if buf is None:
            raise RuntimeError("GradFanIn: the parked gradient is missing (backward order violated)")
        return buf


class _ParkGrad(torch.autograd.Function):
    """Identity whose backward PARKS the incoming gradient in a GradFanIn and reports None: for a map with two consumers of which
    the one that runs its backward LAST is a dense convolution (it adds its data gradient onto the parked one in its epilogue,
    _DenseConvBNAct) and the other one is anything else.  Put on the path to that other consumer."""

    @staticmethod
    def forward(ctx, x, fan):
        ctx.fan = fan
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        fan, ctx.fan = ctx.fan, None
        g = g.contiguous()
        g = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
        if fan.buf is not None:          # another consumer parked first (an unexpected backward order): nothing may be lost
            g = g + fan.buf
        fan.buf = g
        return None, None


def park_grad(x, fan, owned=False):
    """x for the consumer whose backward runs FIRST (the one created later in the forward pass); `fan` was handed to the unit
    that consumes x too and runs its backward last (conv_bn_act(..., fanin=fan), stem_pool(x, fanin=fan)) and is armed if that
    unit can accumulate.  owned: the parked gradient is always a fresh tensor of this module's own backward (a convolution's
    data gradient) - then it may be written in place in an eager backward too."""
    # Only inside a captured segment (dl/engine.py): the convolution adds IN PLACE onto the parked tensor, and there that tensor is
    # the segment's own static output-gradient buffer, refilled before every replay.  In an eager backward the incoming gradient may
    # be a tensor the caller still owns (torch.autograd.backward(outs, grads)): it must not be written to - autograd adds as usual.
    # (DFINE_PARK_EAGER=1: the tests' switch - their gradient tensors are clones they do not look at again)
    if (fan is None or not fan.armed or not x.requires_grad or x.dtype != torch.bfloat16
            or not (owned or (_CAPTURE_POSSIBLE and torch.cuda.is_current_stream_capturing()) or _env("DFINE_PARK_EAGER", "0") == "1")):
        return x
    fan.parking = True
    return _ParkGrad.apply(x, fan)


class _FanOut(torch.autograd.Function):
    """k aliases of one fp32 token-stream tensor, one per consumer.  Backward: the consumers' gradients arrive TOGETHER and are
    summed in one pass (dfine_sum_f32) - autograd adds them pairwise as they arrive (k - 1 launches, 3 (k - 1) tensor passes
    against k + 1).  The decoder's LayerNorm outputs feed 3 - 6 consumers each (self-attention q/k and v inputs, the residual
    path, the deformable attention's query, the gate, the FFN, the heads: ref dfine_decoder.py:214-255,456-476)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.set_materialize_grads(False)
        ctx.k = k
        outs = tuple(x.view_as(x) for _ in range(k))
        memo = getattr(x, "_dfine_bf16", None)
        if memo is not None and memo[0] == x._version:     # the bf16 copy the LayerNorm kernel wrote next to x (_bf16_2d)
            for o in outs:
                o._dfine_bf16 = (o._version, memo[1])
        return outs

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        g0 = live[0]
        ok = all(g.dtype == torch.float32 and g.shape == g0.shape and g.is_contiguous() and g.data_ptr() % 16 == 0 for g in live)
        if not ok or g0.numel() % 4 or len(live) > 8:
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        # (into a new tensor: an incoming gradient may be shared - AddBackward hands ONE tensor to both of its inputs)
        return _hip().sum_f32(live), None


def fan_out(x, k):
    """k aliases of x for k consumers (see _FanOut); x itself k times where the fused sum does not apply."""
    if (k < 2 or not torch.is_tensor(x) or not x.is_cuda or not x.requires_grad or x.dtype != torch.float32 or not torch.is_grad_enabled()):
        return (x,) * k
    return _FanOut.apply(x, k)


class _TakeRowsAndPass(torch.autograd.Function):
    """(t, t.gather(1, ind[..., None].expand(-1, -1, C))) for t = [B, L, C] and DISTINCT indices per image: the encoder memory on
    its way to the decoder's value path, and the rows the query selection picks from it (ref dfine_decoder.py:842-853,888-905).
    ONE node for both consumers, so its backward sees both gradients: the selected rows are added onto the value path's
    gradient in place - the reference zero-fills a [B, L, C] gradient, scatters 300 rows into it and lets autograd add the two
    full-size tensors.  Same sums (one bf16 + bf16 add per element of the selected rows)."""

    @staticmethod
    def forward(ctx, t, ind):
        ctx.save_for_backward(ind)
        ctx.shape = t.shape
        ctx.set_materialize_grads(False)
        return t.view_as(t), t.gather(dim=1, index=ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))

    @staticmethod
    def backward(ctx, g_all, g_rows):
        (ind,) = ctx.saved_tensors
        if g_rows is None:
            return g_all, None
        idx = ind.unsqueeze(-1).expand(-1, -1, ctx.shape[-1])
        if g_all is None:
            return g_rows.new_zeros(ctx.shape).scatter_add_(1, idx, g_rows), None
        # Written in place only when g_all is provably this pass's own tensor - the value path's gradient fresh out of the shared
        # deformable-attention accumulator (a view of it: same address).  Autograd does not promise an incoming gradient is
        # unshared (AddBackward and view backward hand one tensor to several nodes): anything else is added out of place.
        own = g_all.is_contiguous() and _OWNED_VALUE_GRAD[0] is not None and _OWNED_VALUE_GRAD[0] == g_all.data_ptr()
        _OWNED_VALUE_GRAD[0] = None
        if own:
            return g_all.scatter_add_(1, idx, g_rows.to(g_all.dtype)), None
        return g_all.scatter_add(1, idx, g_rows.to(g_all.dtype)), None


def take_rows_and_pass(t, ind):
    """(t for its other consumer, rows ind [B, K] of t [B, L, C]); indices distinct per image.  See _TakeRowsAndPass."""
    if t.is_cuda and t.requires_grad and torch.is_grad_enabled() and _env("DFINE_GRAD_FANIN", "1") == "1":
        return _TakeRowsAndPass.apply(t, ind)
    return t, t.gather(dim=1, index=ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))


class _ChainUse:
    """Marks a GradFanIn handed to a part-wise convolution as the CONSUMER side of a chain (its `fans` are the parking side)."""
    __slots__ = ("fan",)

    def __init__(self, fan):
        self.fan = fan


class _FanSlice(torch.autograd.Function):
    """x[:, lo:lo + n] for the consumers of a chain GradFanIn.  Backward: when the map's whole-tensor consumer parked its gradient
    (the part-wise convolution that reads the map as one of its parts) and the slice's consumers added theirs onto channels
    lo..lo + n of it, that buffer IS the map's gradient: returned as it stands.  Otherwise the plain slice gradient."""

    @staticmethod
    def forward(ctx, x, lo, n, fan):
        ctx.fan, ctx.lo, ctx.n, ctx.shape = fan, lo, n, x.shape
        ctx.set_materialize_grads(False)
        return x.view_as(x) if (lo == 0 and n == x.shape[1]) else x[:, lo:lo + n]

    @staticmethod
    def backward(ctx, g):
        fan, ctx.fan = ctx.fan, None
        if fan.parking and fan.armed:
            if fan.pending != 0:
                raise RuntimeError("GradFanIn chain: a consumer's backward has not run yet (backward order violated)")
            full = fan.take()
            view = full[:, ctx.lo:ctx.lo + ctx.n]
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)           # a consumer outside the chain returned a gradient of its own: autograd added it to the slice
            return full, None, None, None
        if g is None:
            return None, None, None, None
        if ctx.lo == 0 and ctx.n == ctx.shape[1]:
            return g, None, None, None
        full = g.new_zeros(ctx.shape)
        full[:, ctx.lo:ctx.lo + ctx.n] = g
        return full, None, None, None


def fan_slice(x, lo, n):
    """(x[:, lo:lo + n], fan) for a map x = [B, C, H, W] that feeds a part-wise 1x1 convolution whole (conv_bn_act([.., x, ..],
    fans=[.., fan, ..]): created LAST, so its backward runs first and parks x's gradient) and one or more 1x1 convolutions
    through the channel slice (conv_bn_act(slice, fanin=fan)): RepNCSPELAN4's split and its branch outputs (ref
    hybrid_encoder.py:196-206).  The reference lets autograd add the consumers' gradients, zero-fill the slice's gradient to the
    full width and add again; here every consumer adds onto the parked map in its store epilogue (dfine_conv1x1_seg_accum_bf16)."""
    if not (grad_fanin_enabled(x) and x.requires_grad and x.dtype == torch.bfloat16 and _env("DFINE_FAN_CHAIN", "1") == "1"):
        return (x if (lo == 0 and n == x.shape[1]) else x[:, lo:lo + n]), None
    fan = GradFanIn(chain=True, lo=lo)
    return _FanSlice.apply(x, lo, n, fan), fan


def grad_fanin_enabled(x):
    return (_env("DFINE_GRAD_FANIN", "1") == "1" and torch.is_tensor(x) and x.is_cuda and torch.is_grad_enabled()
            and _env("DFINE_HIP_UNITS", "1") == "1")


def fanin_outer_enabled():
    """The hand-offs across blocks (HG_Block's residual connection, the stage outputs that leave the backbone)."""
    return True


class _DenseConvBNAct(torch.autograd.Function):
    """_DenseConv followed by _BNAct as ONE autograd node (the HIP-only plan): the same kernel calls in the same order, half
    the `Function.apply` / backward-node dispatches for the ~250 conv + BatchNorm units of a step - the forward pass is
    host-bound (tools/host_profile.py: ~44 ms of host work against 39 ms of device work per step)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, lab_scale, lab_bias, running_mean, running_var, act, training, momentum, eps,
                fanin=None, residual=None):
        """fanin: GradFanIn of x (this unit is its later consumer); residual: added to the unit's output in the BatchNorm
        apply pass (a residual connection behind the unit; its gradient is the output's)."""
        hip = _hip()
        x = x.contiguous()
        ks = weight.shape[-1]
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        ctx.fanin = None
        if (fanin is not None and ctx.needs_input_grad[0] and x.dtype == torch.bfloat16
                and hip.conv_epilogue_supported(B, cout, cin, H, W, ks)):      # (the data gradient: channels exchanged)
            fanin.armed = True
            ctx.fanin = fanin
            if fanin.chain:
                fanin.pending += 1
        c = hip.conv_forward_bf16(x, _packed_weights(weight, False), cout, ks)
        y, stats = hip.bn_act_forward(c, gamma, beta, running_mean, running_var, lab_scale, lab_bias, act, training, momentum, eps,
                                      residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(x, weight, c, stats, lab_scale)
        ctx.cfg = (act, training, gamma is not None, lab_scale is not None)
        need = ctx.needs_input_grad
        ctx.wslot = _defer_slot(weight) if need[1] else None
        if ctx.wslot is not None:
            ctx.wslot[0].note_use(ctx.wslot[1][0])
        ctx.slot = None
        if lab_scale is not None and lab_bias is not None and need[4] and need[5]:
            slot = _defer_slot(lab_scale, lab_bias)
            if slot is not None and slot[0].grad_offset(slot[1][1]) == slot[0].grad_offset(slot[1][0]) + 1:
                ctx.slot = slot
                for i in slot[1]:
                    slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x, weight, c, stats, lab_scale = ctx.saved_tensors
        act, training, has_affine, has_lab = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != c.dtype:
            dy = dy.to(c.dtype)
        slot = ctx.slot
        dlab_ptr = slot[0].grad_ptr(slot[1][0]) if slot is not None else None
        dc, dg, db, dlab = hip.bn_act_backward(c, dy, stats, lab_scale, act, training, has_affine, has_lab, dlab_ptr)
        dls = dlb = None
        if slot is not None:
            for i in slot[1]:
                slot[0].use_done(i)
        elif has_lab:
            dls, dlb = dlab[0:1], dlab[1:2]
        ks = weight.shape[-1]
        need = ctx.needs_input_grad
        dx = None
        if need[0]:
            B, cin, H, W = x.shape
            if ctx.fanin is not None and ctx.fanin.parking:    # the other consumer's gradient is parked: add onto it in the epilogue
                fan, ctx.fanin = ctx.fanin, None
                if fan.chain:                                  # one of several: add onto the parked map and leave it there
                    view = fan.buf if fan.buf.shape[1] == cin else fan.buf[:, fan.lo:fan.lo + cin]
                    if view.is_contiguous():
                        hip.conv_accumulate_bf16(dc, _packed_weights(weight, True), view, ks)
                    elif ks == 1:
                        hip.conv1x1_seg_forward((dc,), _packed_weights(weight, True), (view,), accum=True)
                    else:
                        view.add_(hip.conv_forward_bf16(dc, _packed_weights(weight, True), cin, ks))
                    fan.pending -= 1
                    dx = view if fan.pending == 0 else None            # (the last one tells _FanSlice the gradient is there)
                else:
                    dx = hip.conv_accumulate_bf16(dc, _packed_weights(weight, True), fan.take(), ks)
            else:
                dx = hip.conv_forward_bf16(dc, _packed_weights(weight, True), weight.shape[1], ks)
        dw = None
        if need[1]:
            wslot = ctx.wslot
            if wslot is not None:
                ws, meta = hip.conv_wgrad_bf16(x, dc, ks, partials=True)
                wslot[0].defer_wgrad(wslot[1][0], ws, meta)
                wslot[0].use_done(wslot[1][0])
            else:
                dw = hip.conv_wgrad_bf16(x, dc, ks).to(weight.dtype)
        return dx, dw, dg, db, dls, dlb, None, None, None, None, None, None, None, (dy if len(need) > 13 and need[13] else None)


class _InnerCtx:
    """Stand-in for the autograd context of a convolution node run INSIDE another node (_ConvBNActAny)."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _ConvBNActAny(torch.autograd.Function):
    """Any convolution node of this module (depthwise, part-wise 1x1, stem ...) followed by _BNAct as ONE autograd node: the
    inner node's forward / backward run as plain functions on a stand-in context.  Same kernels, one node dispatch less per
    unit and direction (the forward pass is host-bound)."""

    @staticmethod
    def forward(ctx, inner, n, *args):
        conv_args = args[:n]
        gamma, beta, lab_scale, lab_bias, running_mean, running_var, act, training, momentum, eps = args[n:]
        ictx = _InnerCtx()
        ictx.needs_input_grad = ctx.needs_input_grad[2:2 + n]
        c = inner.forward(ictx, *conv_args)
        y, stats = _hip().bn_act_forward(c, gamma, beta, running_mean, running_var, lab_scale, lab_bias, act, training, momentum, eps)
        ctx.save_for_backward(c, stats, lab_scale)
        ctx.inner, ctx.ictx, ctx.n = inner, ictx, n
        ctx.cfg = (act, training, gamma is not None, lab_scale is not None)
        ctx.slot = None
        need = ctx.needs_input_grad
        if lab_scale is not None and lab_bias is not None and need[2 + n + 2] and need[2 + n + 3]:
            slot = _defer_slot(lab_scale, lab_bias)
            if slot is not None and slot[0].grad_offset(slot[1][1]) == slot[0].grad_offset(slot[1][0]) + 1:
                ctx.slot = slot
                for i in slot[1]:
                    slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        c, stats, lab_scale = ctx.saved_tensors
        act, training, has_affine, has_lab = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != c.dtype:
            dy = dy.to(c.dtype)
        slot = ctx.slot
        dlab_ptr = slot[0].grad_ptr(slot[1][0]) if slot is not None else None
        dc, dg, db, dlab = _hip().bn_act_backward(c, dy, stats, lab_scale, act, training, has_affine, has_lab, dlab_ptr)
        dls = dlb = None
        if slot is not None:
            for i in slot[1]:
                slot[0].use_done(i)
        elif has_lab:
            dls, dlb = dlab[0:1], dlab[1:2]
        inner_grads = ctx.inner.backward(ctx.ictx, dc)
        ctx.ictx = None
        return (None, None) + tuple(inner_grads) + (dg, db, dls, dlb, None, None, None, None, None, None)


def _bn_tail_fused(inner, conv_args, bn, a, lab):
    """conv node + BatchNorm tail as one autograd node when the BatchNorm is a plain tracked nn.BatchNorm2d; None otherwise."""
    if not (_FUSE_CONV_BN and type(bn) is nn.BatchNorm2d and bn.track_running_stats and bn.momentum is not None):
        return None
    training = bn.training
    if training:
        if _BN_DEFER:
            _bn_count_deferred(bn)
        else:
            bn.num_batches_tracked.add_(1)
    return _ConvBNActAny.apply(inner, len(conv_args), *conv_args, bn.weight, bn.bias, lab.scale if lab is not None else None,
                               lab.bias if lab is not None else None, bn.running_mean, bn.running_var, a, training, bn.momentum, bn.eps)


class _DualConv(torch.autograd.Function):
    """(conv_a(x), conv_b(x)) for a 3x3 and a 1x1 convolution of the same input (RepVGG unit): one op so that the backward
    pass forms d(x) = dgrad_a(dc_a) + dgrad_b(dc_b) with the second data-gradient kernel accumulating onto the first
    (dfine_conv1x1_accum_bf16) instead of autograd adding two maps."""

    @staticmethod
    def forward(ctx, x, wa, wb):
        hip = _hip()
        x = x.contiguous()
        ca = hip.conv_forward_bf16(x, _packed_weights(wa, False), wa.shape[0], wa.shape[-1])
        cb = hip.conv_forward_bf16(x, _packed_weights(wb, False), wb.shape[0], wb.shape[-1])
        ctx.save_for_backward(x, wa, wb)
        ctx.slots = []
        for i, w in ((1, wa), (2, wb)):
            slot = _defer_slot(w) if (ctx.needs_input_grad[i] and hip.conv_wgrad_supported(x.shape[2], x.shape[3], w.shape[-1])) else None
            if slot is not None:
                slot[0].note_use(slot[1][0])
            ctx.slots.append(slot)
        return ca, cb

    @staticmethod
    def backward(ctx, dca, dcb):
        hip = _hip()
        x, wa, wb = ctx.saved_tensors
        dca = dca.contiguous() if dca.dtype == torch.bfloat16 else dca.to(torch.bfloat16).contiguous()
        dcb = dcb.contiguous() if dcb.dtype == torch.bfloat16 else dcb.to(torch.bfloat16).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = hip.conv_forward_bf16(dca, _packed_weights(wa, True), wa.shape[1], wa.shape[-1])
            if not hip.conv1x1_accumulate(dcb, _packed_weights(wb, True), dx):
                dx = dx + hip.conv_forward_bf16(dcb, _packed_weights(wb, True), wb.shape[1], 1)
        grads = []
        for need, w, dy, slot in ((ctx.needs_input_grad[1], wa, dca, ctx.slots[0]), (ctx.needs_input_grad[2], wb, dcb, ctx.slots[1])):
            g = None
            if need:
                ks = w.shape[-1]
                if slot is not None:
                    ws, meta = hip.conv_wgrad_bf16(x, dy, ks, partials=True)
                    slot[0].defer_wgrad(slot[1][0], ws, meta)
                    slot[0].use_done(slot[1][0])
                else:
                    g = hip.conv_wgrad_bf16(x, dy, ks).to(w.dtype)
            grads.append(g)
        return dx, grads[0], grads[1]


class _DenseConvSeg(torch.autograd.Function):
    """1x1 convolution of torch.cat(xs, dim=1) that never builds the concatenation: the HIP kernels gather the input
    channels from the parts (forward, weight gradient) and scatter the data gradient into one tensor per part
    (csrc/conv.hip: ChanSegs)."""

    @staticmethod
    def forward(ctx, weight, fans, *xs):
        """fans: None, or one GradFanIn / None per part - armed ones receive that part's data gradient in backward (the part's
        other consumer then returns the sum), see GradFanIn."""
        hip = _hip()
        ctx.fans = ctx.chain = None
        if fans is not None and len(fans) == 1 and isinstance(fans[0], _ChainUse):
            f, fans = fans[0].fan, None
            B, cin, H, W = xs[0].shape
            if (len(xs) == 1 and ctx.needs_input_grad[2] and xs[0].dtype == torch.bfloat16
                    and hip.conv_epilogue_supported(B, weight.shape[0], cin, H, W, 1)):
                f.armed = True
                f.pending += 1
                ctx.chain = f
        if fans is not None:
            for i, f in enumerate(fans):
                if f is not None and f.armed and ctx.needs_input_grad[2 + i] and xs[i].dtype == torch.bfloat16:
                    f.parking = True
                    ctx.fans = fans
        xs = tuple(x if hip.is_channel_part(x) else x.contiguous() for x in xs)     # channel slices are read in place
        B, _, H, W = xs[0].shape
        y = torch.empty(B, weight.shape[0], H, W, device=xs[0].device, dtype=torch.bfloat16)
        hip.conv1x1_seg_forward(xs, _packed_weights(weight, False), (y,))
        ctx.save_for_backward(weight, *xs)
        ctx.slot = _defer_slot(weight) if ctx.needs_input_grad[0] else None
        if ctx.slot is not None:
            ctx.slot[0].note_use(ctx.slot[1][0])
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        weight, *xs = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dxs = [None] * len(xs)
        need = ctx.needs_input_grad
        chain, ctx.chain = ctx.chain, None
        if chain is not None and chain.parking:
            # the map's gradient is parked (by the part-wise convolution that reads it whole): add onto this unit's channels of it
            view = chain.buf[:, chain.lo:chain.lo + xs[0].shape[1]]
            hip.conv1x1_seg_forward((dy,), _packed_weights(weight, True), (view,), accum=True)
            chain.pending -= 1
            dxs = [view if chain.pending == 0 else None]       # (the last one tells _FanSlice the gradient is there)
        elif any(need[2:]):
            fans, ctx.fans = ctx.fans, None
            # a part whose GradFanIn ALREADY holds a gradient (HG_Block: the residual connection's, parked before this node ran):
            # this data gradient is added onto it in the store epilogue
            pre = [fans is not None and fans[i] is not None and fans[i].parking and fans[i].buf is not None
                   and fans[i].buf.shape == x.shape and fans[i].buf.is_contiguous() for i, x in enumerate(xs)]
            outs = tuple(fans[i].buf if p else torch.empty(x.shape, device=x.device, dtype=x.dtype) for i, (x, p) in enumerate(zip(xs, pre)))
            hip.conv1x1_seg_forward((dy,), _packed_weights(weight, True), outs, accum=pre if any(pre) else False)
            dxs = [o if n else None for o, n in zip(outs, need[2:])]
            if fans is not None:
                for i, f in enumerate(fans):
                    if f is not None and f.parking and dxs[i] is not None:
                        if not pre[i] and f.buf is not None:            # a parked gradient this launch could not add onto
                            dxs[i] = dxs[i] + f.buf
                        f.buf, dxs[i] = dxs[i], None        # parked for the part's other consumer (GradFanIn)
        dw = None
        if need[0]:
            if ctx.slot is not None:
                ws, meta = hip.conv1x1_seg_wgrad(xs, dy, partials=True)
                ctx.slot[0].defer_wgrad(ctx.slot[1][0], ws, meta)
                ctx.slot[0].use_done(ctx.slot[1][0])
            else:
                dw = hip.conv1x1_seg_wgrad(xs, dy).to(weight.dtype)
        return (dw, None, *dxs)


class _DenseConvMFMA(_DenseConv):
    """Same op with the HIP kernels forced (used by the parity tests)."""

    @staticmethod
    def forward(ctx, x, weight):
        hip = _hip()
        x = x.contiguous()
        ks = weight.shape[-1]
        y = hip.conv_forward_bf16(x, hip.conv_pack_weights(weight.detach().float().contiguous(), False),
                                  weight.shape[0], ks)
        ctx.save_for_backward(x, weight)
        ctx.direct = True            # the weight may be a temporary (channel-padded copy): packed per call, never registered
        return y


def _pad16_conv_ok(conv, x):
    """Dense 1x1 / 3x3 stride-1 'same' convolutions whose channel counts are NOT multiples of 16 (the 21 / 43-channel units of
    D-FINE-n's encoder, expansion 0.34): served by the MFMA kernels on zero-padded channels."""
    k = conv.kernel_size
    return (_env("DFINE_MFMA_CONV", "1") == "1" and conv.groups == 1 and k[0] == k[1] and k[0] in (1, 3)
            and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.bias is None
            and isinstance(conv.padding, tuple) and conv.padding == (k[0] // 2, k[0] // 2)
            and (conv.in_channels % 16 != 0 or conv.out_channels % 16 != 0) and x.dim() == 4
            and ((k[0] == 1 and (x.shape[-1] * x.shape[-2]) % 2 == 0)
                 or (k[0] == 3 and x.shape[-1] % 2 == 0 and x.shape[-1] <= 160))
            and _bf16_autocast())


def _dense_conv_pad16(x, weight):
    """conv(x) with input / output channels zero-padded to multiples of 16 around the MFMA kernels (zero input channels and
    zero weight rows add nothing; the extra output channels are dropped).  The padding / slicing ops carry the gradients."""
    cout, cin = weight.shape[0], weight.shape[1]
    cin_p, cout_p = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
    if cin_p != cin:
        xb = F.pad(xb, (0, 0, 0, 0, 0, cin_p - cin))
    w = weight if (cin_p, cout_p) == (cin, cout) else F.pad(weight, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
    y = _DenseConvMFMA.apply(xb, w)
    return y if cout_p == cout else y[:, :cout].contiguous()


# ---- HGNetv2 stem: direct small-channel convolutions + pad-fused max-pool (csrc/stem.hip) ------------
_STEM_DGRAD_S2 = {(48, 24), (32, 16), (64, 32)}      # (Cin, Cout) of the instantiated 3x3 / stride-2 data gradients


def _packed_stem(weight, mode):
    if _CAPTURE_POSSIBLE and not _CAPTURE_FROZEN_WEIGHTS and torch.cuda.is_current_stream_capturing():
        return _hip().stem_pack_weights(weight.detach().float().contiguous(), mode)     # recorded: the weights change between replays
    key = (id(weight), "stem", mode)
    tag = (_WEIGHT_EPOCH, weight._version, weight.data_ptr())
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is weight:
        return hit[1]
    wp = _hip().stem_pack_weights(weight.detach().float().contiguous(), mode)
    _PACK_CACHE[key] = (tag, wp, weakref.ref(weight))
    return wp


class _StemConv(torch.autograd.Function):
    """Stem convolution (3x3/s2, 2x2/s1 on the bottom/right zero-padded map, 1x1), NCHW bf16."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, pad_br):
        hip = _hip()
        x = x.contiguous()
        cout, cin, ks, _ = weight.shape
        H, W = x.shape[2], x.shape[3]
        if pad_br:
            ho, wo = (H + 1 + 2 * pad - ks) // stride + 1, (W + 1 + 2 * pad - ks) // stride + 1
        else:
            ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = hip.stem_conv(x, _packed_stem(weight, 0), cout, ks, stride, pad, (ho, wo))
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        cout, cin, ks, _ = weight.shape
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if stride == 1:
                dx = hip.stem_conv(dy, _packed_stem(weight, 1), cin, ks, 1, ks - 1 - pad, (x.shape[2], x.shape[3]))
            else:
                dx = hip.stem_dgrad_s2(dy, _packed_stem(weight, 2), cin)
        if ctx.needs_input_grad[1]:
            dw = hip.stem_wgrad(x, dy, ks, stride, pad, side=_stem_wgrad_side(weight)).to(weight.dtype)
        return dx, dw, None, None, None


class _StemConv2(torch.autograd.Function):
    """3x3 / stride-2 stem convolution of torch.cat([xa, xb], 1) that never builds the concatenation (StemBlock: pooled stem1
    map + stem2 branch -> stem3, ref hgnetv2.py:158-165): both tensors are read in place and the data gradient comes back as
    two contiguous tensors (the concatenation costs a 630 MB copy forward and two 315 MB `.contiguous()` copies of the
    sliced gradient backward at D-FINE-m / 640 / bs 32)."""

    @staticmethod
    def forward(ctx, xa, xb, weight, pad):
        hip = _hip()
        xa, xb = xa.contiguous(), xb.contiguous()
        cout, cin, ks, _ = weight.shape
        H, W = xa.shape[2], xa.shape[3]
        ho, wo = (H + 2 * pad - ks) // 2 + 1, (W + 2 * pad - ks) // 2 + 1
        y = hip.stem_conv2(xa, xb, _packed_stem(weight, 0), cout, ks, 2, pad, (ho, wo))
        ctx.save_for_backward(xa, xb, weight)
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        xa, xb, weight = ctx.saved_tensors
        ks = weight.shape[-1]
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dxa = dxb = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxa, dxb = hip.stem_dgrad_s2_2(dy, _packed_stem(weight, 2), xa.shape[1], xb.shape[1])
        if ctx.needs_input_grad[2]:
            dw = hip.stem_wgrad2(xa, xb, dy, ks, 2, ctx.pad, side=_stem_wgrad_side(weight)).to(weight.dtype)
        return dxa, dxb, dw, None


class _StemPool(torch.autograd.Function):
    """MaxPool2d(2, stride 1, ceil_mode) of F.pad(x, (0, 1, 0, 1)) without materialising the padded map."""

    @staticmethod
    def forward(ctx, x, fanin=None):
        """fanin: GradFanIn of x - the pool is created BEFORE x's other consumer (StemBlock), so its backward runs last and adds
        its gradient onto the parked one."""
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.fanin = None
        if fanin is not None and ctx.needs_input_grad[0] and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0:
            fanin.armed = True
            ctx.fanin = fanin
        return _hip().stem_pool_forward(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        fan, ctx.fanin = ctx.fanin, None
        if fan is not None and fan.parking:
            return _hip().stem_pool_backward(x, dy, acc=fan.take()), None
        return _hip().stem_pool_backward(x, dy), None


def stem_fast_path(x):
    """The stem kernels serve CUDA tensors under bf16 autocast (the training / bf16 inference path)."""
    return (x.is_cuda and _env("DFINE_HIP_UNITS", "1") == "1" and _env("DFINE_STEM", "1") == "1"
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def stem_pool(x, fanin=None):
    """pool(F.pad(x, (0,1,0,1))) of StemBlock (ref hgnetv2.py:158-160)."""
    if stem_fast_path(x) and x.dtype == torch.bfloat16:
        return _StemPool.apply(x, fanin)
    return F.max_pool2d(F.pad(x, (0, 1, 0, 1)), kernel_size=2, stride=1, ceil_mode=True)


def _stem_conv_ok(conv, x, pad_br):
    if not (stem_fast_path(x) and x.dim() == 4 and conv.groups == 1 and conv.bias is None and conv.dilation == (1, 1)):
        return False
    k, s = conv.kernel_size, conv.stride
    if k[0] != k[1] or s[0] != s[1] or not isinstance(conv.padding, tuple) or conv.padding[0] != conv.padding[1]:
        return False
    ks, st, pad = k[0], s[0], conv.padding[0]
    cin, cout = conv.in_channels, conv.out_channels
    hip = _hip()
    if not hip.stem_supported(cin, cout, ks, st) or cout > 32:
        return False
    H, W = x.shape[2], x.shape[3]
    if pad_br:
        if not (ks == 2 and st == 1 and pad == 0):
            return False
        wo = W
    else:
        if ks == 2:
            return False
        wo = (W + 2 * pad - ks) // st + 1
    # (the MFMA weight-gradient kernel walks 32-pixel K steps: narrower rows run on zero-padded copies of dy, hip.stem_wgrad;
    # its 8-output lane groups read one clamped x window each, which is only right when no group is partly valid)
    if wo % 8 != 0 or W % (8 * st) != 0 or ks - 1 - pad > 1:
        return False
    if st == 1:
        return hip.stem_supported(cout, cin, ks, 1)     # data gradient = same kernel, channels swapped
    if st == 2:
        return ks == 3 and pad == 1 and H % 2 == 0 and W % 2 == 0 and (
            (cin, cout) in _STEM_DGRAD_S2 or not x.requires_grad)
    return False


def _mfma_conv_ok(conv, x, allow_bias=False):
    """Layers the implicit-GEMM kernels can serve (1x1 / 3x3, stride 1, 'same' padding, bf16 autocast)."""
    k = conv.kernel_size
    return (_env("DFINE_MFMA_CONV", "1") == "1" and conv.groups == 1 and k[0] == k[1] and k[0] in (1, 3)
            and conv.stride == (1, 1) and conv.dilation == (1, 1) and (conv.bias is None or allow_bias)
            and isinstance(conv.padding, tuple) and conv.padding == (k[0] // 2, k[0] // 2)
            and conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0 and x.dim() == 4
            and ((k[0] == 1 and (x.shape[-1] * x.shape[-2]) % 2 == 0)
                 or (k[0] == 3 and x.shape[-1] % 2 == 0 and x.shape[-1] <= 160))
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def _is_depthwise(conv, allow_bias=False):
    return (conv.groups > 1 and conv.groups == conv.in_channels == conv.out_channels
            and conv.kernel_size[0] == conv.kernel_size[1] <= 7 and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and conv.dilation == (1, 1) and (conv.bias is None or allow_bias)
            and isinstance(conv.padding, tuple))


_ROUTES = [0]       # epoch token of the per-module route caches of conv_bn_act (replaced by reload_env)


def _conv_bn_act_residual(x, conv, bn, a, lab, fanin, residual):
    """conv_bn_act(x, ...) + residual as ONE fused dense unit (the add rides in the BatchNorm apply pass), or None when the
    layer is not served that way (then the caller adds)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.is_contiguous() and x.dtype == torch.bfloat16 and a in (None, "relu", "silu", "swish")
            and _env("DFINE_HIP_UNITS", "1") == "1" and _env("DFINE_BN_RESIDUAL", "1") == "1" and not _is_depthwise(conv)
            and _mfma_conv_ok(conv, x) and _FUSE_CONV_BN and type(bn) is nn.BatchNorm2d and bn.track_running_stats
            and bn.momentum is not None and _conv_plan_all_hip(x, conv.weight) and residual.dtype == torch.bfloat16
            and residual.shape == (x.shape[0], conv.out_channels, x.shape[2], x.shape[3]) and _hip().bn_residual_supported(residual)):
        return None
    training = bn.training
    if training:
        if _BN_DEFER:
            _bn_count_deferred(bn)
        else:
            bn.num_batches_tracked.add_(1)
    return _DenseConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, lab.scale if lab is not None else None,
                                 lab.bias if lab is not None else None, bn.running_mean, bn.running_var, a, training,
                                 bn.momentum, bn.eps, fanin, residual)


EVAL_EPILOGUE = True      # conv -> eval-mode BatchNorm / deployed bias -> act [-> LAB] as one launch when no gradient is recorded


def _eval_fold(bn):
    """(scale, shift) fp32 [C] of an eval-mode BatchNorm.  `freeze_eval_affine` stores them on the module for a model whose weights
    no longer change (Torch_model); otherwise one fold launch per call (running statistics and affine are updated in place by
    kernels that do not bump tensor versions: nothing to key a cache on)."""
    hit = bn.__dict__.get("_dfine_fold")
    if hit is not None:
        if hit[0] == _fold_key(bn):
            return hit[1]
        del bn.__dict__["_dfine_fold"]            # weights were loaded / moved since freeze_eval_affine: fold per call again
    return _hip().bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _fold_key(bn):
    """(address, version) of the four tensors a fold is made of: load_state_dict / copy_ / .to() all show in it (the training
    kernels' in-place updates do not - a model that trains must not be frozen)."""
    return tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))


def _lab_pair(lab):
    if lab is None:
        return None
    hit = lab.__dict__.get("_dfine_pair")
    if hit is not None:
        return hit
    return torch.cat([lab.scale.detach().reshape(1), lab.bias.detach().reshape(1)]).float()


def freeze_eval_affine(model):
    """Inference-only models (weights fixed from here on): the eval-mode BatchNorm folds and learnable-affine pairs are computed
    once and kept on the modules, so that a conv -> BN -> act unit is exactly one launch.  Call again after loading new weights."""
    n = 0
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d) or type(m).__name__ == "FrozenBatchNorm2d":
            m.__dict__.pop("_dfine_fold", None)
            if m.running_mean is not None and m.running_mean.is_cuda:
                m.__dict__["_dfine_fold"] = (_fold_key(m), _hip().bn_fold(m.weight, m.bias, m.running_mean, m.running_var, m.eps))
                n += 1
        elif type(m).__name__ == "LearnableAffineBlock":
            m.__dict__.pop("_dfine_pair", None)
            if m.scale.is_cuda:
                m.__dict__["_dfine_pair"] = _lab_pair(m)
    return n


def _eval_unit_ok(bn, a):
    return (EVAL_EPILOGUE and not torch.is_grad_enabled() and not bn.training and a in (None, "relu", "silu", "swish")
            and getattr(bn, "running_mean", None) is not None and getattr(bn, "weight", None) is not None
            and bn.running_mean.dtype == torch.float32 and bn.weight.dtype == torch.float32)


def _conv_eval_affine(xb, conv, scale, shift, a, lab):
    """The one-launch inference unit on a whole bf16 tensor, or None when the shape's kernel has no epilogue."""
    hip = _hip()
    B, cin, H, W = xb.shape
    ks = conv.kernel_size[0]
    if not hip.conv_affine_supported(B, cin, conv.out_channels, H, W, ks):
        return None
    return hip.conv_forward_affine(xb.contiguous(), _packed_weights(conv.weight, False), conv.out_channels, ks, scale, shift,
                                   "silu" if a == "swish" else a, _lab_pair(lab))


def conv_bn_act(x, conv: nn.Conv2d, bn: nn.Module, act: Optional[str], lab: Optional[nn.Module],
                pad_br: bool = False, fanin=None, fans=None, residual=None):
    """conv(bias=False) -> BN (batch stats in training) -> {None, relu, silu} -> scalar affine; the
    building block of HGNetv2 and the HybridEncoder.
    fanin / fans: GradFanIn hand-offs of the data gradient (fanin: this unit is the LATER consumer of x in backward; fans: one
    per part of a list input, this unit being the EARLIER one) - only honoured by the fused HIP units, ignored elsewhere.
    GPU (bf16 autocast): dense 1x1 / 3x3, depthwise and stem convolutions and the whole BN/act/affine tail are HIP kernels;
    fp32 math and CPU tensors take the plain ATen composition below."""
    a = act.lower() if isinstance(act, str) else act
    if residual is not None:
        # only the fused dense unit adds it in its apply pass; every other route: the unit, then a plain add
        y = _conv_bn_act_residual(x, conv, bn, a, lab, fanin, residual)
        if y is not None:
            return y
        return conv_bn_act(x, conv, bn, act, lab, pad_br, fanin, fans) + residual
    if (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and not x.is_contiguous() and conv.kernel_size == (1, 1)
            and x.dtype == torch.bfloat16 and _hip().is_channel_part(x)):
        x = [x]                       # a channel slice of a wider map (RepNCSPELAN4 split): read in place, no .contiguous() copy
        if fanin is not None and fanin.chain:
            fans = [_ChainUse(fanin)]
    if isinstance(x, (list, tuple)):
        # channel-wise concatenation kept as parts: the 1x1 MFMA kernels read them in place
        xs = x
        if (xs[0].is_cuda and not pad_br and _env("DFINE_HIP_UNITS", "1") == "1"
                and _env("DFINE_SEG_CONV", "1") == "1" and a in (None, "relu", "silu", "swish")
                and conv.kernel_size == (1, 1) and len(xs) <= 8 and (xs[0].shape[-1] * xs[0].shape[-2]) % 8 == 0
                and all(t.shape[1] % 8 == 0 for t in xs) and _mfma_conv_ok(conv, xs[0])):
            parts = [t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16) for t in xs]
            if _eval_unit_ok(bn, a):
                hip = _hip()
                B, _, H, W = parts[0].shape
                cin = sum(t.shape[1] for t in parts)
                if hip.conv_affine_supported(B, cin, conv.out_channels, H, W, 1):
                    parts = [t if hip.is_channel_part(t) else t.contiguous() for t in parts]
                    scale, shift = _eval_fold(bn)
                    return hip.conv1x1_seg_forward_affine(parts, _packed_weights(conv.weight, False), conv.out_channels, scale, shift,
                                                          "silu" if a == "swish" else a, _lab_pair(lab))
            y = _bn_tail_fused(_DenseConvSeg, (conv.weight, fans, *parts), bn, a, lab)
            if y is not None:
                return y
            y = _DenseConvSeg.apply(conv.weight, fans, *parts)
            return _bn_tail(y, bn, a, act, lab)
        if (len(xs) == 2 and xs[0].is_cuda and not pad_br and a in (None, "relu", "silu", "swish") and _env("DFINE_HIP_UNITS", "1") == "1"
                and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
                and xs[0].shape[2:] == xs[1].shape[2:] and xs[0].shape[2] % 2 == 0 and xs[0].shape[3] % 2 == 0
                and (conv.in_channels, conv.out_channels) in _STEM_DGRAD_S2 and xs[0].shape[1] + xs[1].shape[1] == conv.in_channels
                and _stem_conv_ok(conv, xs[0], False)):
            xa, xb = (t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16) for t in xs)
            y = _StemConv2.apply(xa, xb, conv.weight, conv.padding[0])
            return _bn_tail(y, bn, a, act, lab)
        x = torch.cat(list(xs), dim=1) if len(xs) > 1 else xs[0]
    if x.is_cuda and a in (None, "relu", "silu", "swish") and _env("DFINE_HIP_UNITS", "1") == "1":
        # Which kernel serves this layer depends on the module and on the input's shape / type only: decided once per
        # (module, input signature) - the predicates below cost ~10 us of attribute look-ups per call, 300 calls per step, and
        # the forward pass is host-bound (tools/host_profile.py).
        routes = conv.__dict__.get("_dfine_routes")
        if routes is None or routes[0] is not _ROUTES:
            routes = conv.__dict__["_dfine_routes"] = (_ROUTES, {})      # per module; a new epoch object after reload_env()
        key = (x.shape, x.dtype, pad_br, x.requires_grad, torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
        route = routes[1].get(key)
        if route is None:
            if _is_depthwise(conv):
                route = 1
            elif not pad_br and _mfma_conv_ok(conv, x):
                route = 2
            elif (not pad_br and conv.kernel_size == (3, 3) and 160 < x.shape[-1] <= 304 and x.shape[-1] % 2 == 0
                  and _mfma_conv_ok(conv, x[..., :_DenseConvWide._halves(x.shape[-1])[1]])):
                route = 3
            elif _stem_conv_ok(conv, x, pad_br):
                route = 4
            elif _f32_conv_ok(conv, x):
                route = 5
            elif not pad_br and _pad16_conv_ok(conv, x):
                route = 6
            else:
                route = 0
            if len(routes[1]) > 64:
                routes[1].clear()
            routes[1][key] = route
        if route == 1:
            if x.dtype == torch.float32 and torch.is_autocast_enabled():
                x = x.to(torch.get_autocast_dtype("cuda"))      # what autocast would do for F.conv2d
                fanin = None
            if (_eval_unit_ok(bn, a) and x.dtype == torch.bfloat16
                    and _hip().dwconv_affine_supported(x, conv.kernel_size[0], conv.stride[0], conv.padding[0])):
                return _hip().dwconv_forward_affine(x.contiguous(), conv.weight.detach().float().contiguous(), conv.stride[0],
                                                    conv.padding[0], *_eval_fold(bn), "silu" if a == "swish" else a, _lab_pair(lab))
            y = _bn_tail_fused(_DepthwiseConv, (x, conv.weight, conv.stride[0], conv.padding[0], fanin), bn, a, lab)
            if y is not None:
                return y
            y = _DepthwiseConv.apply(x, conv.weight, conv.stride[0], conv.padding[0])
        elif route == 2:
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            if _eval_unit_ok(bn, a):
                y = _conv_eval_affine(xb, conv, *_eval_fold(bn), a, lab)
                if y is not None:
                    return y
            if (_FUSE_CONV_BN and type(bn) is nn.BatchNorm2d and bn.track_running_stats and bn.momentum is not None
                    and _conv_plan_all_hip(xb, conv.weight)):
                training = bn.training
                if training:
                    if _BN_DEFER:
                        _bn_count_deferred(bn)
                    else:
                        bn.num_batches_tracked.add_(1)
                return _DenseConvBNAct.apply(xb, conv.weight, bn.weight, bn.bias, lab.scale if lab is not None else None,
                                             lab.bias if lab is not None else None, bn.running_mean, bn.running_var, a, training,
                                             bn.momentum, bn.eps, fanin if xb is x else None)
            y = _DenseConv.apply(xb, conv.weight)
        elif route == 3:
            # maps wider than the kernel's 160-pixel strips (the 240-wide stage of D-FINE-l / x at 960 x 960): two column halves
            y = _DenseConvWide.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv.weight)
        elif route == 4:
            y = _StemConv.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv.weight,
                                conv.stride[0], conv.padding[0], pad_br)
        elif route == 5:
            y = conv_f32(x, conv, pad_br)           # fp32 math (configs[1]): the f32-input MFMA kernels
        elif route == 6:
            y = _dense_conv_pad16(x, conv.weight)
        else:
            if _bf16_autocast():
                _library_fallback(f"{conv} on an input of shape {tuple(x.shape)}")
            y = conv(F.pad(x, (0, 1, 0, 1)) if pad_br else x)
        return _bn_tail(y, bn, a, act, lab)
    y = bn(conv(F.pad(x, (0, 1, 0, 1)) if pad_br else x))
    return _act_lab_torch(y, act, lab)


_UNIT_BN = {}        # (device, channels) -> (ones, zeros): the identity BatchNorm a bias + activation pass is expressed with


def conv_bias_act(x, conv: nn.Conv2d, act: Optional[str], residual=None):
    """act(conv(x) + bias) [+ residual] for a convolution whose BatchNorm was folded away by `model.deploy()`
    (ConvNormLayer_fuse.conv_bn_fused, the re-parameterised RepVGG 3x3: ref hybrid_encoder.py:47-79,123-156).
    CUDA under bf16 autocast: the same MFMA implicit-GEMM kernels as the training form, then ONE pass of the fused
    BatchNorm / activation kernel with unit statistics (scale 1, shift = bias).  Otherwise the ATen composition."""
    a = act.lower() if isinstance(act, str) else act
    if isinstance(x, (list, tuple)) and len(x) > 1:
        # a channel-wise concatenation kept as parts (the FPN / PAN inputs of the CSP layers): inference reads them in place through
        # the part-wise 1x1 kernel with the bias + activation epilogue - no concatenated copy (8 x 78 us per batch-32 forward)
        xs = x
        if (EVAL_EPILOGUE and not torch.is_grad_enabled() and xs[0].is_cuda and a in (None, "relu", "silu", "swish")
                and conv.kernel_size == (1, 1) and conv.bias is not None and conv.bias.dtype == torch.float32 and len(xs) <= 8
                and (xs[0].shape[-1] * xs[0].shape[-2]) % 8 == 0 and all(t.shape[1] % 8 == 0 for t in xs)
                and _bf16_autocast() and _mfma_conv_ok(conv, xs[0], allow_bias=True)):
            hip = _hip()
            B, _, H, W = xs[0].shape
            cin = sum(t.shape[1] for t in xs)
            if cin == conv.in_channels and hip.conv_affine_supported(B, cin, conv.out_channels, H, W, 1):
                parts = [t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16) for t in xs]
                parts = [t if hip.is_channel_part(t) else t.contiguous() for t in parts]
                c = conv.out_channels
                unit = _UNIT_BN.get((parts[0].device, c))
                if unit is None:
                    unit = _UNIT_BN[(parts[0].device, c)] = (torch.ones(c, device=parts[0].device), torch.zeros(c, device=parts[0].device))
                y = hip.conv1x1_seg_forward_affine(parts, _packed_weights(conv.weight, False), c, unit[0], conv.bias.detach(),
                                                   "silu" if a == "swish" else a, None)
                return y if residual is None else y + residual
    if isinstance(x, (list, tuple)):
        x = torch.cat(list(x), dim=1) if len(x) > 1 else x[0]
    autocast16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    dense = _mfma_conv_ok(conv, x, allow_bias=True)
    if (x.is_cuda and a in (None, "relu", "silu", "swish") and _env("DFINE_HIP_UNITS", "1") == "1" and conv.bias is not None
            and x.dim() == 4 and (dense or (autocast16 and _is_depthwise(conv, allow_bias=True)))):
        c = conv.out_channels
        unit = _UNIT_BN.get((x.device, c))
        if unit is None:
            unit = _UNIT_BN[(x.device, c)] = (torch.ones(c, device=x.device), torch.zeros(c, device=x.device))
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        if dense and EVAL_EPILOGUE and not torch.is_grad_enabled() and conv.bias.dtype == torch.float32:
            y = _conv_eval_affine(xb, conv, unit[0], conv.bias.detach(), a, None)          # bias + activation in the store phase
            if y is not None:
                return y if residual is None else y + residual
        y = _DenseConv.apply(xb, conv.weight) if dense else _DepthwiseConv.apply(xb, conv.weight, conv.stride[0], conv.padding[0])
        y = _BNAct.apply(y, unit[0], conv.bias, None, None, unit[1], unit[0], a, False, 0.0, 0.0)
        return y if residual is None else y + residual
    y = _act_lab_torch(conv(x), act, None)
    return y if residual is None else y + residual


class _BN2Act(torch.autograd.Function):
    """act(BN_1(c1) + BN_2(c2)) [+ residual], batch statistics (HIP: bnact.hip, RepVGG unit)."""

    @staticmethod
    def forward(ctx, c1, c2, residual, g1, b1, rm1, rv1, g2, b2, rm2, rv2, act, mom1, eps1, mom2, eps2):
        c1, c2 = c1.contiguous(), c2.contiguous()
        res = None if residual is None else residual.contiguous()
        y, saved = _hip().bn2_act_forward(c1, c2, res, (g1, b1, rm1, rv1, mom1, eps1), (g2, b2, rm2, rv2, mom2, eps2), act)
        ctx.save_for_backward(c1, c2, saved)
        ctx.cfg = (act, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, c2, saved = ctx.saved_tensors
        act, has_res = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != c1.dtype:
            dy = dy.to(c1.dtype)
        need = ctx.needs_input_grad
        d1, d2, dg1, db1, dg2, db2 = _hip().bn2_act_backward(c1, c2, dy, saved, act, (need[3] or need[4], need[7] or need[8]))
        return (d1, d2, dy if has_res else None, dg1, db1, None, None, dg2, db2, None, None, None, None, None, None, None)


def _bn_trainable(bn):
    return (isinstance(bn, nn.BatchNorm2d) and bn.training and bn.track_running_stats and bn.momentum is not None
            and bn.affine)


class _DualConvBN2Act(torch.autograd.Function):
    """_DualConv followed by _BN2Act as ONE autograd node (the RepVGG unit of the encoder in training form)."""

    @staticmethod
    def forward(ctx, x, wa, wb, residual, g1, b1, rm1, rv1, g2, b2, rm2, rv2, act, mom1, eps1, mom2, eps2):
        ictx = _InnerCtx()
        ictx.needs_input_grad = ctx.needs_input_grad[:3]
        c1, c2 = _DualConv.forward(ictx, x, wa, wb)
        res = None if residual is None else residual.contiguous()
        y, saved = _hip().bn2_act_forward(c1, c2, res, (g1, b1, rm1, rv1, mom1, eps1), (g2, b2, rm2, rv2, mom2, eps2), act)
        ctx.save_for_backward(c1, c2, saved)
        ctx.ictx = ictx
        ctx.cfg = (act, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, c2, saved = ctx.saved_tensors
        act, has_res = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != c1.dtype:
            dy = dy.to(c1.dtype)
        need = ctx.needs_input_grad
        d1, d2, dg1, db1, dg2, db2 = _hip().bn2_act_backward(c1, c2, dy, saved, act, (need[4] or need[5], need[8] or need[9]))
        dx, dwa, dwb = _DualConv.backward(ctx.ictx, d1, d2)
        ctx.ictx = None
        return (dx, dwa, dwb, dy if has_res else None, dg1, db1, None, None, dg2, db2, None, None, None, None, None, None, None)


def repvgg_unit(x, conv1: nn.Conv2d, bn1, conv2: nn.Conv2d, bn2, act: Optional[str], residual=None):
    """act(bn1(conv1(x)) + bn2(conv2(x))) [+ residual]: the RepVGG block of the hybrid encoder in training form (ref
    hybrid_encoder.py:106-156) and CSPLayer's residual (hybrid_encoder.py:209-239).  On the GPU in bf16 training the two
    BatchNorms, the add, the activation and the residual add are ONE apply pass (csrc/bnact.hip: dfine_bn2_act_*);
    everything else composes the unit from conv_bn_act."""
    a = act.lower() if isinstance(act, str) else act
    if x.is_cuda and _FUSE_CONV_BN:
        # the fully fused form (dual conv + two BatchNorms + add + activation [+ residual] as one node), decided once per
        # (module, input signature, BatchNorm mode)
        routes = conv1.__dict__.get("_dfine_rep_routes")
        if routes is None or routes[0] is not _ROUTES:
            routes = conv1.__dict__["_dfine_rep_routes"] = (_ROUTES, {})
        key = (x.shape, x.dtype, a, bn1.training, bn2.training, None if residual is None else (residual.shape, residual.dtype),
               torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
        ok = routes[1].get(key)
        if ok is None:
            ok = bool(a in (None, "relu", "silu", "swish") and _env("DFINE_HIP_UNITS", "1") == "1" and _env("DFINE_BN2", "1") == "1"
                      and _bn_trainable(bn1) and _bn_trainable(bn2) and type(bn1) is nn.BatchNorm2d and type(bn2) is nn.BatchNorm2d
                      and _mfma_conv_ok(conv1, x) and _mfma_conv_ok(conv2, x)
                      and conv1.kernel_size == (3, 3) and conv2.kernel_size == (1, 1) and conv1.out_channels == conv2.out_channels
                      and _env("DFINE_DUAL_CONV", "1") == "1"
                      and _hip().bn2_supported(torch.empty(x.shape[0], conv1.out_channels, x.shape[2], x.shape[3], device="meta",
                                                           dtype=torch.bfloat16))
                      and (residual is None or (residual.shape == (x.shape[0], conv1.out_channels, x.shape[2], x.shape[3])
                                                and residual.dtype == torch.bfloat16)))
            if len(routes[1]) > 64:
                routes[1].clear()
            routes[1][key] = ok
        if ok:
            for bn in (bn1, bn2):
                if _BN_DEFER:
                    _bn_count_deferred(bn)
                else:
                    bn.num_batches_tracked.add_(1)
            return _DualConvBN2Act.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv1.weight, conv2.weight, residual,
                                         bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias,
                                         bn2.running_mean, bn2.running_var, a, bn1.momentum, bn1.eps, bn2.momentum, bn2.eps)
    if (x.is_cuda and a in (None, "relu", "silu", "swish") and _env("DFINE_HIP_UNITS", "1") == "1"
            and _env("DFINE_BN2", "1") == "1" and _bn_trainable(bn1) and _bn_trainable(bn2)
            and _mfma_conv_ok(conv1, x) and _mfma_conv_ok(conv2, x)):
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        if conv1.kernel_size == (3, 3) and conv2.kernel_size == (1, 1) and _env("DFINE_DUAL_CONV", "1") == "1":
            c1, c2 = _DualConv.apply(xb, conv1.weight, conv2.weight)
        else:
            c1 = _DenseConv.apply(xb, conv1.weight)
            c2 = _DenseConv.apply(xb, conv2.weight)
        if c1.shape == c2.shape and _hip().bn2_supported(c1) and (residual is None or (
                residual.shape == c1.shape and residual.dtype == torch.bfloat16)):
            for bn in (bn1, bn2):
                if _BN_DEFER:
                    _bn_count_deferred(bn)
                else:
                    bn.num_batches_tracked.add_(1)
            return _BN2Act.apply(c1, c2, residual, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                                 bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, a,
                                 bn1.momentum, bn1.eps, bn2.momentum, bn2.eps)
        y = _bn_tail(c1, bn1, None, None, None) + _bn_tail(c2, bn2, None, None, None)
    else:
        y = conv_bn_act(x, conv1, bn1, None, None) + conv_bn_act(x, conv2, bn2, None, None)
    y = _act_lab_torch(y, act, None)
    return y if residual is None else y + residual


def _bn_tail(y, bn, a, act, lab):
    """BatchNorm (+ activation + learnable affine) of a conv output on the GPU: one fused HIP op (bnact.hip)."""
    if isinstance(bn, nn.BatchNorm2d) and bn.track_running_stats and bn.momentum is not None:
        training = bn.training
        if training:
            if _BN_DEFER:
                _bn_count_deferred(bn)
            else:
                bn.num_batches_tracked.add_(1)
        return _BNAct.apply(y, bn.weight, bn.bias, lab.scale if lab is not None else None,
                            lab.bias if lab is not None else None, bn.running_mean, bn.running_var,
                            a, training, bn.momentum, bn.eps)
    if (not isinstance(bn, nn.modules.batchnorm._BatchNorm) and torch.is_tensor(getattr(bn, "running_var", None))
            and torch.is_tensor(getattr(bn, "weight", None)) and hasattr(bn, "eps")):
        # FrozenBatchNorm2d: buffers only, always "eval" statistics
        return _BNAct.apply(y, bn.weight, bn.bias, lab.scale if lab is not None else None,
                            lab.bias if lab is not None else None, bn.running_mean, bn.running_var,
                            a, False, 0.0, bn.eps)
    return _act_lab_torch(bn(y), act, lab)


def _act_lab_torch(y, act, lab):
    if act is not None:
        a = act.lower()
        if a == "relu":
            y = F.relu(y)
        elif a in ("silu", "swish"):
            y = F.silu(y)
        elif a == "gelu":
            y = F.gelu(y)
        else:
            raise RuntimeError(f"unsupported activation {act}")
    if lab is not None:
        y = lab.scale * y + lab.bias
    return y


# =============================================================================================
# A5/A6  residual / gate + LayerNorm of the token streams (csrc/lnfused.hip)
# =============================================================================================
_LN_DIMS = (64, 128, 256, 384, 512, 1024)


LN_DEFER = True       # (tools/ab_step.py kernels.LN_DEFER)


class _LNFused(torch.autograd.Function):
    """mode 0: LN(a + b); mode 1: LN(clamp(a + b)); mode 2: LN(sigmoid(g[:, :D]) * a + sigmoid(g[:, D:]) * b)."""

    @staticmethod
    def forward(ctx, mode, a, b, gate, weight, bias, eps, clampv):
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        gate = None if gate is None else gate.contiguous()
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            # the stream stays fp32 (residual path, as in the reference under autocast); the GEMMs that read it next take
            # the bf16 copy written in the same pass (picked up by _bf16_2d) instead of casting it themselves
            y, mean, rstd, y16 = _hip().ln_fused_forward(mode, a, b, gate, weight, bias, eps, clampv, with_bf16=True)
            y._dfine_bf16 = (y._version, y16)
        else:
            y, mean, rstd = _hip().ln_fused_forward(mode, a, b, gate, weight, bias, eps, clampv)
        ctx.save_for_backward(a, b, gate, weight, mean, rstd)
        ctx.cfg = (mode, clampv, bias is not None)
        # the affine gradients ride in the step's deferred split reduction (one launch for every conv / linear / LayerNorm
        # parameter) instead of a column-sum launch per LayerNorm: 13 per D-FINE-m step
        ctx.slot = None
        if LN_DEFER and bias is not None and ctx.needs_input_grad[4] and ctx.needs_input_grad[5]:
            ctx.slot = _defer_slot(weight, bias)
            if ctx.slot is not None:
                for i in ctx.slot[1]:
                    ctx.slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, gate, weight, mean, rstd = ctx.saved_tensors
        mode, clampv, has_bias = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != torch.float32:
            dy = dy.float()
        need = ctx.needs_input_grad
        if ctx.slot is not None:
            fused, idx = ctx.slot
            da, db, dg, ws, blocks = _hip().ln_fused_backward(mode, a, b, gate, weight, mean, rstd, dy, clampv, need[1], need[2],
                                                              need[3], True, partials=True)
            D = weight.numel()
            meta = (blocks, D, 1, 1, 2 * D, 1)               # [blocks][2 D] rows: a "bias" row of the deferred reduction
            fused.defer_wgrad(idx[0], ws, meta)
            fused.defer_wgrad(idx[1], ws, meta, ws_offset=D)
            for i in idx:
                fused.use_done(i)
            return None, da, db, dg, None, None, None, None
        da, db, dg, dw, dbias = _hip().ln_fused_backward(mode, a, b, gate, weight, mean, rstd, dy, clampv, need[1],
                                                         need[2], need[3], need[4] or (has_bias and need[5]))
        return None, da, db, dg, (dw if need[4] else None), (dbias if has_bias and need[5] else None), None, None


def _ln_fast(x, norm):
    return (x.is_cuda and _env("DFINE_LN_FUSED", "1") == "1" and isinstance(norm, nn.LayerNorm)
            and norm.elementwise_affine and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] in _LN_DIMS
            and x.shape[-1] == norm.normalized_shape[0] and x.dtype in (torch.float32, torch.bfloat16))


def add_layer_norm(x, branch, norm: nn.LayerNorm, clamp: Optional[float] = None):
    """norm(x + branch) or norm((x + branch).clamp(-clamp, clamp)) - the residual LayerNorms of the decoder /
    encoder layers (ref dfine_decoder.py:238-255, hybrid_encoder.py:243-280).  One HIP pass on CUDA tensors."""
    if _ln_fast(x, norm) and branch.dtype in (torch.float32, torch.bfloat16) and branch.shape == x.shape:
        return _LNFused.apply(0 if clamp is None else 1, x, branch, None, norm.weight, norm.bias, norm.eps,
                              0.0 if clamp is None else float(clamp))
    z = x + branch
    if clamp is not None:
        z = z.clamp(min=-clamp, max=clamp)
    return norm(z)


def layer_norm(x, norm: nn.LayerNorm):
    """Plain LayerNorm over the last dim (enc_output.norm, ref dfine_decoder.py:615-621); HIP on CUDA tensors."""
    if _ln_fast(x, norm):
        return _LNFused.apply(0, x, None, None, norm.weight, norm.bias, norm.eps, 0.0)
    return norm(x)


def linear_module(mod: nn.Linear, x, act=None):
    """`mod(x)` (an nn.Linear kept for its state-dict keys) through the HIP GEMM."""
    return linear(x, mod.weight, mod.bias, act=act)


def gate_layer_norm(gate_logits, x1, x2, norm: nn.LayerNorm):
    """norm(sigmoid(g)[..., :D] * x1 + sigmoid(g)[..., D:] * x2), g = Gate.gate([x1, x2]) (ref dfine_decoder.py:258-271)."""
    if (_ln_fast(x1, norm) and x2.shape == x1.shape and gate_logits.shape[-1] == 2 * x1.shape[-1]
            and x2.dtype in (torch.float32, torch.bfloat16) and gate_logits.dtype in (torch.float32, torch.bfloat16)):
        return _LNFused.apply(2, x1, x2, gate_logits, norm.weight, norm.bias, norm.eps, 0.0)
    g1, g2 = torch.sigmoid(gate_logits).chunk(2, dim=-1)
    return norm(g1 * x1 + g2 * x2)


def _bf16_2d(x):
    """[..., K] activation -> contiguous bf16 [M, K] (what the GEMM / weight-gradient kernels read).
    The cast of an fp32 tensor is remembered on the tensor (the decoder's fp32 token stream feeds several linears each -
    score / box heads, value projection, FFN: 22 casts of the same few [B, Q, 256] tensors per step)."""
    if x.dtype == torch.bfloat16:
        return x.reshape(-1, x.shape[-1]).contiguous()
    memo = getattr(x, "_dfine_bf16", None)
    if memo is not None and memo[0] == x._version:
        return memo[1]
    xb = x.to(torch.bfloat16)
    xb = xb.reshape(-1, xb.shape[-1]).contiguous()
    if x.dtype == torch.float32 and x.is_contiguous():
        x._dfine_bf16 = (x._version, xb)
    return xb


def _f32_vec(b):
    return None if b is None else (b.detach() if b.dtype == torch.float32 and b.is_contiguous() else b.detach().float().contiguous())


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b) for the token streams of encoder / decoder, all three GEMMs on the HIP MFMA kernels:
    forward and data gradient = linear_act_kernel (csrc/gemm.hip; the activation, the bias and - for the data gradient -
    nothing else are fused in), weight + bias gradient = the split-K linear_wgrad_kernel (csrc/conv.hip).
    `act`: 0 none, 1 relu, 2 gelu, 3 silu.  bf16 operands / fp32 accumulate (what bf16 autocast gives F.linear)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        hip = _hip()
        x2d = _bf16_2d(x)
        wb = bf16_param(weight)
        b32 = _f32_vec(bias)
        need_grad = any(ctx.needs_input_grad[:3])
        ref = None
        n = weight.shape[0]
        # the result is a fresh base tensor (not a view): callers may modify it in place (ReLU(inplace=True) in the heads)
        y = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.bfloat16)
        if act >= 2 and need_grad:          # GELU / SiLU: backward needs the pre-activation
            ref = hip.linear_act(x2d, wb, b32, 0, out=y.view(-1, n))
            ref, y = y, hip.act_forward(y, act)
        else:
            hip.linear_act(x2d, wb, b32, act, out=y.view(-1, n))
            if act == 1:
                ref = y
        ctx.save_for_backward(x2d, weight, ref)
        ctx.meta = (act, x.dtype, x.shape, bias is not None)
        ctx.slot = None
        if ctx.needs_input_grad[1] and (bias is None or ctx.needs_input_grad[2]):
            ctx.slot = _defer_slot(weight) if bias is None else _defer_slot(weight, bias)
            if ctx.slot is not None:
                for i in ctx.slot[1]:
                    ctx.slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x2d, weight, ref = ctx.saved_tensors
        act, xdt, xshape, has_bias = ctx.meta
        d2 = _bf16_2d(dy)
        if act:
            d2 = hip.act_backward(d2, ref.view(-1, ref.shape[-1]), act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.linear_act(d2, bf16_param_t(weight), None, 0, out_f32=xdt == torch.float32).view(xshape)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.slot is not None:
            fused, idx = ctx.slot
            ws, wmeta, bmeta, boff = hip.linear_wgrad_partials(x2d, d2)
            fused.defer_wgrad(idx[0], ws, wmeta)
            if has_bias:
                fused.defer_wgrad(idx[1], ws, bmeta, ws_offset=boff)
            for i in idx:
                fused.use_done(i)
        elif ctx.needs_input_grad[1] or want_db:
            res = hip.linear_wgrad_bf16(x2d, d2, with_bias=want_db)
            dw, db = res if want_db else (res, None)
            if dw.dtype != weight.dtype:
                dw = dw.to(weight.dtype)
        return dx, dw, db, None


class _MLPRelu(torch.autograd.Function):
    """Linear -> ReLU -> Linear [-> ReLU -> Linear ...] (MLP heads, the decoder's FFN: ref dfine_decoder.py:33-46,214-231) as ONE
    autograd node.  Forward: the _LinearAct kernels, layer by layer.  Backward: the data gradient of layer i + 1 carries the
    ReLU backward of layer i in its store epilogue (dfine_linear_dgrad_relu: masked by the saved output) - the reference, and
    one _LinearAct per layer, run threshold_backward as a pass of its own between the two GEMMs (24 launches per D-FINE-m step);
    the weight gradients are registered for the grouped launch as before.  Same values as the per-layer composition."""

    @staticmethod
    def forward(ctx, x, n, *params):
        hip = _hip()
        ws, bs = params[:n], params[n:]
        h = _bf16_2d(x)
        saved = [h]
        for i, (w, b) in enumerate(zip(ws, bs)):
            y = torch.empty(h.shape[0], w.shape[0], device=x.device, dtype=torch.bfloat16)
            hip.linear_act(h, bf16_param(w), _f32_vec(b), 1 if i + 1 < n else 0, out=y)
            h = y
            if i + 1 < n:
                saved.append(h)
        ctx.save_for_backward(*saved, *ws)
        ctx.meta = (n, x.dtype, x.shape, tuple(b is not None for b in bs))
        ctx.slots = []
        need = ctx.needs_input_grad
        for i, (w, b) in enumerate(zip(ws, bs)):
            slot = None
            if need[2 + i] and (b is None or need[2 + n + i]):
                slot = _defer_slot(w) if b is None else _defer_slot(w, b)
                if slot is not None:
                    for k in slot[1]:
                        slot[0].note_use(k)
            ctx.slots.append(slot)
        return h.view(*x.shape[:-1], ws[-1].shape[0])

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        n, xdt, xshape, has_bias = ctx.meta
        saved = ctx.saved_tensors
        acts, ws = saved[:n], saved[n:]
        need = ctx.needs_input_grad
        d2 = _bf16_2d(dy)
        dws, dbs = [None] * n, [None] * n
        dx = None
        for i in range(n - 1, -1, -1):
            w, xin, slot = ws[i], acts[i], ctx.slots[i]
            want_db = has_bias[i] and need[2 + n + i]
            # the data gradient first: the weight gradient below is only registered (grouped launch at the flush)
            nxt = None
            if i > 0:
                nxt = hip.linear_dgrad_relu(d2, bf16_param_t(w), xin)
            elif need[0]:
                dx = hip.linear_act(d2, bf16_param_t(w), None, 0, out_f32=xdt == torch.float32).view(xshape)
            if slot is not None:
                fused, idx = slot
                wsp, wmeta, bmeta, boff = hip.linear_wgrad_partials(xin, d2)
                fused.defer_wgrad(idx[0], wsp, wmeta)
                if has_bias[i]:
                    fused.defer_wgrad(idx[1], wsp, bmeta, ws_offset=boff)
                for k in idx:
                    fused.use_done(k)
            elif need[2 + i] or want_db:
                res = hip.linear_wgrad_bf16(xin, d2, with_bias=want_db)
                dws[i], dbs[i] = res if want_db else (res, None)
                if dws[i].dtype != w.dtype:
                    dws[i] = dws[i].to(w.dtype)
            d2 = nxt
        return (dx, None, *dws, *dbs)


MLP_FUSED = True      # (tools/ab_step.py kernels.MLP_FUSED flips it for an in-process A/B)


def mlp_relu(x, layers):
    """layers[-1](relu(... relu(layers[0](x)))) for nn.Linear `layers` (>= 2).  CUDA under bf16 autocast: one autograd node with the
    ReLU backward fused into the data-gradient GEMMs (_MLPRelu); otherwise the per-layer `linear` composition."""
    ws = [m.weight for m in layers]
    if (MLP_FUSED and len(layers) >= 2 and x.numel() > 0 and all(_hip_linear_ok(x, w) for w in ws)
            and all(w.shape[0] % 8 == 0 for w in ws[:-1]) and torch.is_grad_enabled()):
        return _MLPRelu.apply(x, len(layers), *ws, *[m.bias for m in layers])
    for m in layers[:-1]:
        x = linear(x, m.weight, m.bias, act="relu")
    return linear(x, layers[-1].weight, layers[-1].bias)


class _ClampPos(torch.autograd.Function):
    """clamp(x, -10, 10) of the decoder's query position embedding (ref dfine_decoder.py:466) as one launch each way: ATen's
    ClampBackward1 is two compares, a logical and a multiply - 4 launches per decoder layer in the host-paced stretch of the
    step."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _hip().act_forward(x, 4)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _hip().act_backward(dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous(), x, 4)


def clamp_pos(x):
    """x.clamp(min=-10, max=10); bf16 CUDA tensors on the HIP element-wise kernels."""
    if x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 8 == 0 and x.numel() > 0 \
            and torch.is_grad_enabled() and x.requires_grad:
        return _ClampPos.apply(x)
    return x.clamp(min=-10, max=10)


_ACT_CODES = {None: 0, "none": 0, "relu": 1, "gelu": 2, "silu": 3, "swish": 3}


def _act_code(act):
    if act is None or isinstance(act, str):
        return _ACT_CODES.get(act.lower() if isinstance(act, str) else None)
    if isinstance(act, nn.Identity):
        return 0
    if isinstance(act, nn.ReLU):
        return 1
    if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none":
        return 2
    if isinstance(act, nn.SiLU):
        return 3
    return None


def _hip_linear_ok(x, weight):
    return (x.is_cuda and weight.dim() == 2 and weight.dtype in (torch.float32, torch.bfloat16)
            and _env("DFINE_HIP_LINEAR", "1") == "1"
            and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
                                               and x.dtype == torch.float32)))


class _LinearF32(torch.autograd.Function):
    """fp32 nn.Linear (config #2, no autocast) on the f32-input MFMA GEMM (csrc/gemm_f32.hip): forward x W^T + b (ReLU fused),
    data gradient dY (W^T)^T, weight gradient dY^T (x^T)^T as split partial products over the token rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        hip = _hip()
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        w = weight if weight.stride(-1) == 1 else weight.contiguous()
        y = hip.gemm_f32_nt(x2, w, bias, act=1 if relu else 0)
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.cfg = (relu, bias is not None, x.shape)
        # the split partial products of the weight gradient (and the bias column sums as a one-split row) ride in the step's
        # deferred reduction instead of a sum launch + an accumulate launch per parameter (~120 per D-FINE-s step)
        ctx.slot = None
        if ctx.needs_input_grad[1] and w is weight and (bias is None or ctx.needs_input_grad[2]):
            ctx.slot = _defer_slot(weight) if bias is None else _defer_slot(weight, bias)
            if ctx.slot is not None:
                for i in ctx.slot[1]:
                    ctx.slot[0].note_use(i)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x2, w, y = ctx.saved_tensors
        relu, has_bias, xshape = ctx.cfg
        N = w.shape[0]
        d2 = dy.reshape(-1, N)
        need = ctx.needs_input_grad
        bias_part = None
        if ctx.slot is not None and has_bias and hip.colsum_f32_ok(d2):
            bias_part, d2 = hip.colsum_f32(d2, y if relu else None)        # ReLU mask + bias column sums in one pass
        elif relu:
            d2 = d2 * (y > 0)
        elif not d2.is_contiguous():
            d2 = d2.contiguous()
        dx = dw = db = None
        if need[0]:
            dx = hip.gemm_f32(d2, w, b_kmajor=True).view(xshape)                 # dY [M, N] . W [N, K]: W read in place
        if need[1]:
            M = x2.shape[0]
            splits = max(1, min(64, M // 256))
            part = hip.gemm_f32(d2, x2, a_kmajor=True, b_kmajor=True, splits=splits)   # dY^T x: both read in place, token rows = K
            if ctx.slot is not None:
                fused, idx = ctx.slot
                K = x2.shape[1]
                fused.defer_wgrad(idx[0], part.view(-1), (part.shape[0] if part.dim() == 3 else 1, N, K, 1, N, K))
                if has_bias and bias_part is not None:
                    fused.defer_wgrad(idx[1], bias_part.view(-1), (bias_part.shape[0], N, 1, 1, N, 1))
                elif has_bias:
                    fused.defer_wgrad(idx[1], d2.sum(0), (1, N, 1, 1, N, 1))
                for i in idx:
                    fused.use_done(i)
                return dx, None, None, None
            dw = part.sum(0) if part.dim() == 3 else part
        if has_bias and need[2]:
            db = d2.sum(0)
        return dx, dw, db, None


class _BmmF32(torch.autograd.Function):
    """alpha * a @ b^T (b_kmajor False: b [Z, N, K]) or alpha * a @ b (b_kmajor True: b [Z, K, N]) for fp32 batches - the two
    products of an attention head and their four gradients on the f32 GEMM kernel, every operand read in place."""

    @staticmethod
    def forward(ctx, a, b, alpha, b_kmajor):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.cfg = (alpha, b_kmajor)
        return _hip().gemm_f32(a, b, b_kmajor=b_kmajor, alpha=alpha)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        alpha, bkm = ctx.cfg
        hip = _hip()
        dc = dc.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:          # dA [M, K] = dC [M, N] . B ([N, K]: K-major for this product) or . B^T ([K, N]: NT)
            da = hip.gemm_f32(dc, b, b_kmajor=not bkm, alpha=alpha)
        if ctx.needs_input_grad[1]:
            if bkm:                          # B [K, N]: dB = A^T dC
                db = hip.gemm_f32(a, dc, a_kmajor=True, b_kmajor=True, alpha=alpha)
            else:                            # B [N, K]: dB = dC^T A
                db = hip.gemm_f32(dc, a, a_kmajor=True, b_kmajor=True, alpha=alpha)
        return da, db, None, None


def _f32_gemm_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and x.numel() > 0
            and not torch.is_autocast_enabled() and _env("DFINE_F32_GEMM", "1") == "1")


def attention_f32(q, k, v, allowed=None):
    """softmax(q k^T / sqrt(d) [masked]) v for fp32 [B, H, L, d] tensors: both products on the f32 MFMA GEMM, the softmax an
    ATen element-wise / reduction kernel (no rocBLAS, no SDPA library kernel).  `allowed`: bool [L, L], True = may attend."""
    B, H, L, d = q.shape
    s = _BmmF32.apply(q.reshape(B * H, L, d), k.reshape(B * H, -1, d), float(d) ** -0.5, False)
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = _BmmF32.apply(p, v.reshape(B * H, -1, d), 1.0, True)
    return o.view(B, H, L, d)


def linear(x, weight, bias=None, act=None):
    """act(nn.Linear(x)) for [..., K] activations; `act`: None / "relu" / "gelu" / "silu" or the nn module.
    CUDA under bf16 autocast (or bf16 inputs): HIP GEMM with the activation fused.  Otherwise (fp32 math, CPU) the ATen
    composition."""
    code = _act_code(act)
    if code is not None and _hip_linear_ok(x, weight) and x.numel() > 0:
        return _LinearAct.apply(x, weight, bias, code)
    if _f32_gemm_ok(x, weight):
        y = _LinearF32.apply(x, weight, bias, code == 1)       # fp32 math (configs[1]): the f32-input MFMA GEMM, ReLU fused
        if code == 1:
            return y
    else:
        if x.is_cuda and x.numel() > 0 and _bf16_autocast() and _env("DFINE_HIP_LINEAR", "1") == "1":
            _library_fallback(f"linear {tuple(weight.shape)} on {x.dtype} activations {tuple(x.shape)}")
        y = F.linear(x, weight, bias)
    if act is None or code == 0:
        return y
    if isinstance(act, nn.Module):
        return act(y)
    return {1: F.relu, 2: F.gelu, 3: F.silu}[code](y)


class _MHA(torch.autograd.Function):
    """Packed multi-head self-attention block (q = k input `qk`, v input `value`), every piece a HIP kernel:
    in-projections + out-projection = linear_act_kernel, softmax(QK^T / sqrt(d) + mask) V = attn_fwd_kernel, backward =
    attn_bwd_dq / attn_bwd_dkdv + the same GEMM kernels + the split-K weight gradients written straight into one
    [3E, E] gradient buffer (no slice / pad / cat kernels around the packed in_proj parameters)."""

    @staticmethod
    def forward(ctx, qk, value, in_w, in_b, out_w, out_b, num_heads, mask_u8):
        hip = _hip()
        B, L, E = qk.shape
        qk2, v2 = _bf16_2d(qk), _bf16_2d(value)
        wb = bf16_param(in_w)                                   # [3E, E]
        b32 = _f32_vec(in_b)
        qkp = hip.linear_act(qk2, wb[: 2 * E], b32[: 2 * E]).view(B, L, 2 * E)
        vp = hip.linear_act(v2, wb[2 * E:], b32[2 * E:]).view(B, L, E)
        o, lse2 = hip.attn_forward(qkp[..., :E], qkp[..., E:], vp, num_heads, mask_u8)
        y = hip.linear_act(o.view(B * L, E), bf16_param(out_w), _f32_vec(out_b)).view(B, L, E)
        ctx.save_for_backward(qk2, v2, qkp, vp, o, lse2, in_w, out_w, mask_u8)
        ctx.meta = (num_heads, qk.dtype, value.dtype)
        ctx.slot = _defer_slot(in_w, in_b, out_w, out_b) if all(ctx.needs_input_grad[2:6]) else None
        if ctx.slot is not None:
            for i in ctx.slot[1]:
                ctx.slot[0].note_use(i)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        qk2, v2, qkp, vp, o, lse2, in_w, out_w, mask_u8 = ctx.saved_tensors
        num_heads, qdt, vdt = ctx.meta
        B, L, E = o.shape
        d2 = _bf16_2d(dy)
        do = hip.linear_act(d2, bf16_param_t(out_w)).view(B, L, E)
        slot = ctx.slot
        d_out_w = d_out_b = None
        if slot is not None:
            fused, (i_w, i_b, o_w, o_b) = slot
            ws, wmeta, bmeta, boff = hip.linear_wgrad_partials(o.view(B * L, E), d2)
            fused.defer_wgrad(o_w, ws, wmeta)
            fused.defer_wgrad(o_b, ws, bmeta, ws_offset=boff)
        else:
            d_out_w, d_out_b = hip.linear_wgrad_bf16(o.view(B * L, E), d2, with_bias=True)
        dqk = torch.empty(B, L, 2 * E, device=o.device, dtype=torch.bfloat16)
        dv = torch.empty(B, L, E, device=o.device, dtype=torch.bfloat16)
        hip.attn_backward(qkp[..., :E], qkp[..., E:], vp, o, do, lse2, num_heads, dqk[..., :E], dqk[..., E:], dv, mask_u8)
        wt = bf16_param_t(in_w)                                 # [E, 3E]
        d_qk_in = hip.linear_act(dqk.view(B * L, 2 * E), wt[:, : 2 * E], out_f32=qdt == torch.float32).view(B, L, E)
        d_v_in = hip.linear_act(dv.view(B * L, E), wt[:, 2 * E:], out_f32=vdt == torch.float32).view(B, L, E)
        if slot is not None:
            ws, wmeta, bmeta, boff = hip.linear_wgrad_partials(qk2, dqk.view(B * L, 2 * E))
            fused.defer_wgrad(i_w, ws, wmeta)                                   # rows [0, 2E) of in_proj_weight
            fused.defer_wgrad(i_b, ws, bmeta, ws_offset=boff)
            ws, wmeta, bmeta, boff = hip.linear_wgrad_partials(v2, dv.view(B * L, E))
            fused.defer_wgrad(i_w, ws, wmeta, dst_offset=2 * E * E)             # rows [2E, 3E)
            fused.defer_wgrad(i_b, ws, bmeta, ws_offset=boff, dst_offset=2 * E)
            for i in (i_w, i_b, o_w, o_b):
                fused.use_done(i)
            return d_qk_in, d_v_in, None, None, None, None, None, None
        d_in_w = torch.empty(3 * E, E, device=o.device, dtype=torch.float32)
        d_in_b = torch.empty(3 * E, device=o.device, dtype=torch.float32)
        hip.linear_wgrad_bf16(qk2, dqk.view(B * L, 2 * E), with_bias=True, dw=d_in_w[: 2 * E], db=d_in_b[: 2 * E])
        hip.linear_wgrad_bf16(v2, dv.view(B * L, E), with_bias=True, dw=d_in_w[2 * E:], db=d_in_b[2 * E:])
        return d_qk_in, d_v_in, d_in_w, d_in_b, d_out_w, d_out_b, None, None


def self_attention(qk, value, in_w, in_b, out_w, out_b, num_heads: int, attn_mask=None):
    """Packed-QKV multi-head attention, q = k = `qk` (content+position), v = `value`;
    boolean `attn_mask` [L, L], True = blocked (ref hybrid_encoder.py:243-290, dfine_decoder.py:200,233-255)."""
    b, l, e = qk.shape
    hd = e // num_heads
    if (hd in (8, 16, 24, 32, 40, 48, 56, 64) and e % 8 == 0 and in_b is not None and out_b is not None and _hip_linear_ok(qk, in_w)
            and _env("DFINE_HIP_ATTN", "1") == "1" and (attn_mask is None or (attn_mask.dtype == torch.bool and attn_mask.shape == (l, l)))):
        m8 = None if attn_mask is None else attn_mask.contiguous().view(torch.uint8)
        return _MHA.apply(qk, value, in_w, in_b, out_w, out_b, num_heads, m8)
    q, k = linear(qk, in_w[: 2 * e], in_b[: 2 * e]).chunk(2, dim=-1)
    v = linear(value, in_w[2 * e:], in_b[2 * e:])
    q, k, v = (t.reshape(b, l, num_heads, hd).transpose(1, 2) for t in (q, k, v))
    mask = None if attn_mask is None else ~attn_mask
    if _f32_gemm_ok(q, in_w):
        o = attention_f32(q, k, v, mask)
    elif q.is_cuda and _bf16_autocast() and _env("DFINE_HIP_ATTN", "1") == "1" and _env("DFINE_F32_GEMM", "1") == "1":
        # head dims the bf16 attention kernels do not take (48: the AIFI layer of D-FINE-x): both products on the f32-input
        # MFMA GEMM (csrc/gemm_f32.hip) - the build's own kernels, a fraction of a millisecond for the one layer concerned
        with torch.autocast("cuda", enabled=False):
            o = attention_f32(q.float(), k.float(), v.float(), mask).to(q.dtype)
    else:
        if q.is_cuda and _bf16_autocast() and _env("DFINE_HIP_ATTN", "1") == "1":
            _library_fallback(f"self-attention with head dim {hd} (kernels: multiples of 8 up to 64)")
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    return linear(o.transpose(1, 2).reshape(b, l, e), out_w, out_b)


# BatchNorm bookkeeping: nn.BatchNorm2d bumps `num_batches_tracked` (an int64 device scalar) every forward -
# 133 tiny launches per step.  A train loop may defer them and flush once per step with one multi-tensor add.
_BN_DEFER = False
_BN_PENDING = {}


def _bn_count_deferred(bn):
    """One more training-mode forward of `bn`: remembered, added to its counter by flush_bn_counters.  Keyed by the COUNTER tensor's
    id - the entry holds that tensor, so the id cannot be handed to another object while the entry exists (keyed by the module,
    a dropped model's pending counts were inherited by whatever module got its address next)."""
    buf = bn.num_batches_tracked
    ent = _BN_PENDING.get(id(buf))
    _BN_PENDING[id(buf)] = (buf, 1 if ent is None else ent[1] + 1)


def defer_bn_counters(flag=True):
    global _BN_DEFER
    _BN_DEFER = flag


def flush_bn_counters():
    if _BN_PENDING:
        bufs = [b for b, _ in _BN_PENDING.values()]
        counts = [n for _, n in _BN_PENDING.values()]
        if len(set(counts)) == 1:
            torch._foreach_add_(bufs, counts[0])
        else:
            for b, n in zip(bufs, counts):
                b.add_(n)
        _BN_PENDING.clear()


def topk_anchors(logits: torch.Tensor, k: int) -> torch.Tensor:
    """A3 query selection: indices [B, k] of the anchors with the largest max-over-classes logit,
    descending.  GPU: one fused HIP kernel (csrc/topk.hip) instead of a max-reduce + torch.topk."""
    b, q, c = logits.shape
    if logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and logits.stride(2) == 1 \
            and q <= 16384 and k <= min(q, 1024):
        return _hip().topk_anchors(logits, k)
    return torch.topk(logits.max(-1).values, k, dim=-1).indices


def detection_topk(logits: torch.Tensor, boxes: torch.Tensor, k: int, height: int, width: int, to_round: bool = True):
    """A18 post-processor core (ref export.py:61-100): the k best (query, class) pairs of every image by sigmoid score ->
    (labels [B,k] i64, query index [B,k] i64, absolute xyxy boxes [B,k,4] f32, scores [B,k] f32), descending.
    One HIP kernel (csrc/postproc.hip) for k <= 4096 (any class count); beyond that the reference's own composition on the device."""
    if not logits.is_cuda:
        return _backend_for_cpu("detection_topk")(logits, boxes, k, height, width, to_round)
    if k > 4096:                                       # outside the kernel's sort network: export.py:61-100 restated with device ops
        b, q, c = logits.shape
        scores, index = torch.topk(torch.sigmoid(logits.float()).flatten(1), k, dim=-1)
        query = index // c
        bx = boxes.float() * boxes.new_tensor([width, height, width, height], dtype=torch.float32)
        xy0, xy1 = bx[..., :2] - bx[..., 2:] / 2, bx[..., :2] + bx[..., 2:] / 2
        lim = bx.new_tensor([width, height])
        if to_round:
            xy0, xy1 = torch.clamp(torch.floor(xy0), min=1), torch.minimum(torch.ceil(xy1), lim - 1)
        else:
            xy0, xy1 = torch.clamp(xy0, min=0), torch.minimum(xy1, lim)
        out = torch.cat([xy0, xy1], -1).gather(1, query.unsqueeze(-1).expand(-1, -1, 4))
        return index - query * c, query, out, scores
    return _hip().postprocess(logits, boxes, k, height, width, to_round)


def preprocess_frames(frames: torch.Tensor, out_hw, resized_hw, top_left=(0, 0), pad_value: int = 114, dtype=torch.float32):
    """(f1) uint8 BGR frames [B, H, W, 3] -> network input [B, 3, Ho, Wo] (RGB / 255): bilinear resize to `resized_hw` with
    OpenCV's 8-bit arithmetic, placed at `top_left`, padded with `pad_value` (ref torch_model.py:240-298,378-418).  One HIP kernel."""
    if not frames.is_cuda:
        return _backend_for_cpu("preprocess_frames")(frames, out_hw, resized_hw, top_left, pad_value, dtype)
    return _hip().preprocess_u8(frames, out_hw, resized_hw, top_left, pad_value, dtype)


def topk_indices(score: torch.Tensor, k: int) -> torch.Tensor:
    """Indices of the k largest entries per row, descending.  [ATen plumbing]"""
    return torch.topk(score, k, dim=-1).indices


# =============================================================================================
# A10 / A15  segmentation head: GroupNorm, bilinear resize, mask logits, mask losses, mask costs (csrc/mask.hip)
# =============================================================================================
def _mask_hip_ok(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and _env("DFINE_HIP_MASK", "1") == "1"


class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        x = x.contiguous()
        g32, b32 = _f32_vec(gamma), _f32_vec(beta)
        y, stat = _hip().groupnorm_forward(x, g32, b32, groups, eps, relu)
        ctx.save_for_backward(x, g32, b32, stat)
        ctx.cfg = (groups, relu, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, b32, stat = ctx.saved_tensors
        groups, relu, pdt = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx, dg, db = _hip().groupnorm_backward(x, dy, g32, b32, stat, groups, relu)
        return dx, dg.to(pdt), db.to(pdt), None, None, None


def group_norm_act(x, gn: nn.GroupNorm, relu: bool = False):
    """[relu](gn(x)) - MaskDecoder's normalisation (ref dfine_decoder.py:316-370).  CUDA: one statistics pass + one apply pass
    (csrc/mask.hip), in the dtype of x (under autocast ATen's group_norm would return fp32 maps: 2x the traffic of the
    1/4-resolution maps for nothing the following bf16 convolution keeps)."""
    if _mask_hip_ok(x) and x.dim() == 4 and gn.affine:
        return _GroupNormAct.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, bool(relu))
    y = gn(x)
    return F.relu(y) if relu else y


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, base, out_hw):
        x = x.contiguous()
        if base is not None:
            base = base.contiguous()
            if base.dtype != x.dtype:
                base = base.to(x.dtype)
        ctx.in_hw = tuple(x.shape[-2:])
        ctx.has_base = base is not None
        return _hip().bilinear_forward(x, out_hw, base)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = _hip().bilinear_backward(dy, ctx.in_hw) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.has_base else None), None


def bilinear_resize(x, size, base=None):
    """[base +] F.interpolate(x, size=size, mode="bilinear", align_corners=False): forward and (gather) backward in one HIP
    kernel each; `base` rides in the same pass (the upsample-sum of MaskDecoder's lateral maps)."""
    size = (int(size[0]), int(size[1]))
    if _mask_hip_ok(x) and x.dim() >= 3 and (base is None or base.shape[-2:] == size):
        return _Bilinear.apply(x, base, size)
    y = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    return y if base is None else base + y


_SLAB_PERM = {}


def _pack_rows_1x1(w, device):
    """[B, N, K] float -> bf16 [B, NP, KP] in the k order the 1x1 MFMA kernels read (conv.hip: tr_slab_channel)."""
    B, n, k = w.shape
    NP, KP = (n + 15) // 16 * 16, (k + 31) // 32 * 32
    perm = _SLAB_PERM.get((KP, device))
    if perm is None:
        idx = torch.arange(KP)
        kk = idx % 32
        g, e = kk // 8, kk % 8
        ch = torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4))
        perm = _SLAB_PERM[(KP, device)] = ((idx // 32) * 32 + ch).to(device)
    out = torch.zeros(B, NP, KP, device=device, dtype=torch.bfloat16)
    out[:, :n, :k] = w
    return out[:, :, perm].contiguous()


class _MaskLogits(torch.autograd.Function):
    """einsum("bqc,bchw->bqhw") on the MFMA 1x1 kernel with one weight set per image (conv.hip: dfine_conv1x1_bw_bf16)."""

    @staticmethod
    def forward(ctx, emb, feat):
        hip = _hip()
        feat = feat.contiguous()
        B, Q, C = emb.shape
        y = hip.conv1x1_batched_weights(feat, _pack_rows_1x1(emb.detach(), emb.device), Q)
        ctx.save_for_backward(emb, feat)
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        emb, feat = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        B, Q, C = emb.shape
        d_emb = d_feat = None
        if ctx.needs_input_grad[1]:
            d_feat = hip.conv1x1_batched_weights(dy, _pack_rows_1x1(emb.detach().transpose(1, 2), emb.device), C)
        if ctx.needs_input_grad[0]:
            d_emb = torch.stack([hip.conv_wgrad_bf16(feat[b:b + 1], dy[b:b + 1], 1).view(Q, C) for b in range(B)]).to(emb.dtype)
        return d_emb, d_feat


def mask_logits(emb, feat):
    """emb [B, Q, C] (already scaled), feat [B, C, H, W] -> [B, Q, H, W] (ref dfine_decoder.py:925-932)."""
    if (_mask_hip_ok(feat) and feat.dtype == torch.bfloat16 and emb.shape[1] % 4 == 0 and emb.shape[2] % 4 == 0
            and (feat.shape[2] * feat.shape[3]) % 8 == 0 and _hip().conv_wgrad_supported(feat.shape[2], feat.shape[3], 1)):
        return _MaskLogits.apply(emb, feat)
    return torch.einsum("bqc,bchw->bqhw", emb.to(feat.dtype) if emb.dtype != feat.dtype else emb, feat)


class _DenseConvWide(torch.autograd.Function):
    """3x3 convolution of a map wider than the implicit-GEMM kernel's 160-pixel strips (the 240-wide 1/4-resolution maps of
    the segmentation head at 960 x 960): two overlapping column halves through the same kernels, each half's own output
    columns kept.  Forward / data gradient: halves with a 8-column apron; weight gradient: the sum of the halves' gradients with
    the other half's output columns zeroed in dY."""

    @staticmethod
    def _halves(W):
        half = W // 2
        wl = (half + 8 + 7) // 8 * 8                   # left: columns [0, wl), right: columns [W - wl, W)
        return half, wl

    @staticmethod
    def _run(hip, x, w2, cout):
        B, _, H, W = x.shape
        half, wl = _DenseConvWide._halves(W)
        yl = hip.conv_forward_bf16(x[..., :wl].contiguous(), w2, cout, 3)
        yr = hip.conv_forward_bf16(x[..., W - wl:].contiguous(), w2, cout, 3)
        return torch.cat([yl[..., :half], yr[..., wl - (W - half):]], dim=-1)

    @staticmethod
    def forward(ctx, x, weight):
        hip = _hip()
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        return _DenseConvWide._run(hip, x, _packed_weights(weight, False), weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x, weight = ctx.saved_tensors
        dy = dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous()
        W = x.shape[-1]
        half, wl = _DenseConvWide._halves(W)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _DenseConvWide._run(hip, dy, _packed_weights(weight, True), weight.shape[1])
        if ctx.needs_input_grad[1]:
            dyl = dy[..., :wl].clone()
            dyl[..., half:] = 0
            dyr = dy[..., W - wl:].clone()
            dyr[..., :wl - (W - half)] = 0
            dw = (hip.conv_wgrad_bf16(x[..., :wl].contiguous(), dyl, 3) + hip.conv_wgrad_bf16(x[..., W - wl:].contiguous(), dyr, 3)).to(weight.dtype)
        return dx, dw


def conv_plain(x, conv: nn.Conv2d):
    """conv(x) for a bias-free 1x1 / 3x3 stride-1 convolution without normalisation behind it (MaskDecoder's lateral / fusion /
    up convolutions): the MFMA kernels under bf16 autocast, ATen otherwise."""
    if x.is_cuda and _env("DFINE_HIP_UNITS", "1") == "1":
        if _mfma_conv_ok(conv, x):
            return _DenseConv.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv.weight)
        W = x.shape[-1]
        if conv.kernel_size == (3, 3) and 160 < W <= 304 and W % 2 == 0:
            probe = x[..., :_DenseConvWide._halves(W)[1]]          # the halves the wide form would run
            if _mfma_conv_ok(conv, probe):
                return _DenseConvWide.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv.weight)
        if _bf16_autocast():
            _library_fallback(f"{conv} on an input of shape {tuple(x.shape)}")
    return conv(x)


class _MaskLosses(torch.autograd.Function):
    """-> tensor[2] = (cropped BCE, cropped Dice) of the matched masks (ref dfine_criterion.py:335-450)."""

    @staticmethod
    def forward(ctx, pm, plan_b, plan_q, plan_t, tgt, boxes, eps):
        hip = _hip()
        pm = pm.contiguous()
        tgt, boxes = tgt.float().contiguous(), boxes.float().contiguous()
        sums = hip.mask_loss_sums(pm, plan_b, plan_q, plan_t, tgt, boxes)
        bx = boxes if plan_t is None else boxes[plan_t]
        area = ((bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])).clamp(min=1.0)
        num, den = 2.0 * sums[:, 1] + eps, sums[:, 2] + sums[:, 3] + eps
        out = torch.stack([(sums[:, 0] / area).mean(), (1.0 - num / den).mean()])
        ctx.save_for_backward(pm, plan_b, plan_q, plan_t, tgt, boxes, area, num, den)
        return out

    @staticmethod
    def backward(ctx, g):
        pm, plan_b, plan_q, plan_t, tgt, boxes, area, num, den = ctx.saved_tensors
        m = float(plan_b.numel())
        coef = torch.stack([g[0] / (m * area), -2.0 * g[1] / (m * den), g[1] * num / (m * den * den)], dim=1).float().contiguous()
        return _hip().mask_loss_grad(pm, plan_b, plan_q, plan_t, tgt, boxes, coef), None, None, None, None, None, None


def mask_losses(pm, plan_b, plan_q, tgt, boxes, eps=1e-6, plan_t=None):
    """pm [B, Q, H, W] mask logits, (plan_b, plan_q) [M] int64 matched (image, query); tgt [rows, H, W] in [0, 1] and boxes
    [rows, 4] in mask pixels, row of match m = plan_t[m] (targets of the whole batch, prepared once per step) or m
    -> (loss_mask_bce, loss_mask_dice); read in place (no gathered copies), gradient written in one pass."""
    out = _MaskLosses.apply(pm, plan_b, plan_q, plan_t, tgt, boxes, float(eps))
    return out[0], out[1]


def mask_cost_sums(pm, gt, toff, q, tmax, alpha, gamma):
    return _hip().mask_cost_sums(pm.contiguous(), gt.float().contiguous(), toff, q, tmax, alpha, gamma)


# =============================================================================================
# A1 / A2 in fp32 (configs[1]): dense convolutions on the f32-input matrix cores (csrc/conv_f32.hip)
# =============================================================================================
def _packed_f32(weight, dgrad):
    if _CAPTURE_POSSIBLE and not _CAPTURE_FROZEN_WEIGHTS and torch.cuda.is_current_stream_capturing():
        return _hip().conv_f32_pack_weights(weight.detach().float().contiguous(), dgrad)
    key = (id(weight), "f32", dgrad)
    tag = (_WEIGHT_EPOCH, weight._version, weight.data_ptr())
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is weight:
        return hit[1]
    wp = _hip().conv_f32_pack_weights(weight.detach().float().contiguous(), dgrad)
    _PACK_CACHE[key] = (tag, wp, weakref.ref(weight))
    return wp


class _DenseConvF32(torch.autograd.Function):
    """conv2d(x, weight, stride, padding (pt, pl) top / left, output size `out_hw`) in exact fp32 on the MFMA units; the data
    gradient is the same kernel on the transposed + flipped packing (stride 2: on the zero-upsampled gradient)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pt, pl, out_hw):
        hip = _hip()
        ks = weight.shape[-1]
        one = ks == 1 and stride == 1 and pt == 0 and pl == 0
        if not (one and hip.planes_dense(x)):            # (the 1x1 GEMM reads a channel slice of a concatenation in place)
            x = x.contiguous()
        if one:
            y = hip.conv1x1_f32(x, weight.view(weight.shape[0], weight.shape[1]))      # y[b] = W x[b]: the fp32 GEMM kernel
        else:
            y = hip.conv_f32_forward(x, _packed_f32(weight, False), weight.shape[0], ks, stride, pt, pl, out_hw)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pt, pl)
        ctx.slot = _defer_slot(weight) if ctx.needs_input_grad[1] else None
        if ctx.slot is not None:
            ctx.slot[0].note_use(ctx.slot[1][0])
        return y

    @staticmethod
    def backward(ctx, dy):
        hip = _hip()
        x, weight = ctx.saved_tensors
        stride, pt, pl = ctx.cfg
        ks = weight.shape[-1]
        dy = dy.float()
        if not (ks == 1 and stride == 1 and pt == 0 and pl == 0 and hip.planes_dense(dy)):   # (a slice of a concatenation's gradient)
            dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0] and ks == 1 and stride == 1 and pt == 0 and pl == 0:
            dx = hip.conv1x1_f32(dy, weight.view(weight.shape[0], weight.shape[1]).t().contiguous())      # dx[b] = W^T dy[b]
        elif ctx.needs_input_grad[0]:
            src = dy if stride == 1 else hip.upsample2_zero(dy, (2 * dy.shape[2] - 1, 2 * dy.shape[3] - 1))
            dx = hip.conv_f32_forward(src, _packed_f32(weight, True), weight.shape[1], ks, 1, ks - 1 - pt, ks - 1 - pl, tuple(x.shape[2:]))
        if ctx.needs_input_grad[1]:
            if ctx.slot is not None:
                ws, meta = hip.conv_f32_wgrad(x, dy, ks, stride, pt, pl, partials=True)
                ctx.slot[0].defer_wgrad(ctx.slot[1][0], ws, meta)
                ctx.slot[0].use_done(ctx.slot[1][0])
            else:
                dw = hip.conv_f32_wgrad(x, dy, ks, stride, pt, pl).to(weight.dtype)
        return dx, dw, None, None, None, None


def _f32_conv_ok(conv, x):
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    return (_env("DFINE_F32_CONV", "1") == "1" and x.dim() == 4 and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and conv.weight.dtype == torch.float32 and conv.groups == 1 and conv.bias is None and conv.dilation == (1, 1)
            and k[0] == k[1] <= 3 and s[0] == s[1] <= 2 and isinstance(p, tuple) and p[0] == p[1] <= k[0] - 1)


def conv_f32(x, conv: nn.Conv2d, pad_br: bool = False):
    """conv(x) (or conv(F.pad(x, (0, 1, 0, 1))) for the stem's 2x2 convolutions) in fp32 on the HIP kernel."""
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    hi, wi = x.shape[2] + int(pad_br), x.shape[3] + int(pad_br)
    ho, wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
    return _DenseConvF32.apply(x, conv.weight, s, p, p, (ho, wo))
