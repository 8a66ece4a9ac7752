"""ctypes binding of libdfine_hip.so (C ABI in include/dfine_hip.h) + tensor-level launchers.

Importing this module FAILS LOUDLY when the library is missing or does not export a declared
symbol - there is no fallback path.  Every launcher passes `tensor.data_ptr()` and torch's
current HIP stream, allocates outputs/workspaces with torch (device memory plumbing only) and
raises RuntimeError on a non-zero status.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFINE_HIP_LIB") or os.path.join(_HERE, "csrc", "libdfine_hip.so")     # (override: A/B of two builds on one box)
ABI_VERSION = 3

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first "
        "(`python -m custom_d_fine_amd.csrc.build` or `python -c 'import __graft_entry__ as g; g.build()'`)")

_lib = ctypes.CDLL(LIB_PATH)

_P, _I, _F, _L = c_void_p, c_int, c_float, c_int64
_SIGNATURES = {
    "dfine_abi_version": (c_int, []),
    "dfine_last_error": (ctypes.c_char_p, []),
    "dfine_msda_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dfine_msda_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dfine_msda_fused_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P]),
    "dfine_msda_fused_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P]),
    "dfine_msda_fused_bwd_acc": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _I, _P, _F, _P]),
    "dfine_cast_scaled_acc": (c_int, [_P, _P, _I, _I, _L, _P, _P]),
    "dfine_cast_f32_to_bf16": (c_int, [_P, _P, _L, _P]),
    "dfine_match_ws_bytes": (_L, [_I, _I, _I, _I]),
    "dfine_match": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P]),
    "dfine_lsap": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_dwconv_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_dwconv_bwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_dwconv_s2_dgrad_acc": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_bn_ws_floats": (_L, [_I, _I, _I]),
    "dfine_bn2_supported": (c_int, [_I, _I, _I]),
    "dfine_bn2_ws_floats": (_L, [_I, _I, _I]),
    "dfine_bn2_act_fwd": (c_int, [_P] * 14 + [_I, _I, _I, _I, _F, _F, _F, _F, _P]),
    "dfine_bn2_act_bwd": (c_int, [_P] * 11 + [_I, _I, _I, _I, _P]),
    "dfine_bn_act_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "dfine_bn_residual_once": (c_int, [_P]),
    "dfine_head_losses": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _P, _I,
                                   _P, _P, _P, _I, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_head_losses_dev": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _P, _I,
                                       _P, _P, _P, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_criterion_plans_supported": (c_int, [_I, _I, _I]),
    "dfine_criterion_plans_ws_ints": (_L, [_I, _I, _I]),
    "dfine_criterion_plans": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "dfine_criterion_scales": (c_int, [_P, _I, _P, _P, _I, _P, _P]),
    "dfine_head_grads_scale": (c_int, [_P, _P, _L, _P, _P, _L, _P, _P, _L, _I, _P]),
    "dfine_head_losses_prezeroed_once": (c_int, []),
    "dfine_grad_sqnorm": (c_int, [_P, _L, _F, _P, _P]),
    "dfine_grad_sqnorm_ws_floats": (_L, []),
    "dfine_adamw_ema_step": (c_int, [_P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _I, _F, _F, _F, _P]),
    "dfine_ema_update": (c_int, [_P, _P, _L, _F, _P]),
    "dfine_multi_copy_f32": (c_int, [_P, _I, _P, _P]),
    "dfine_multi_add_f32": (c_int, [_P, _I, _P, _P]),
    "dfine_sum_f32": (c_int, [_P, _I, _P, _L, _P]),
    "dfine_multi_cast_bf16": (c_int, [_P, _I, _P]),
    "dfine_conv_packed_elems": (_L, [_I, _I, _I, _I]),
    "dfine_conv_pack_weights": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "dfine_conv_pack_weights_multi": (c_int, [_P, _I, _P]),
    "dfine_maps_tokens_bf16": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_upsample2_nearest_bf16": (c_int, [_P, _P, c_int64, _I, _I, _I, _P]),
    "dfine_embedding_bwd": (c_int, [_P, _P, _I, _P, c_int64, _I, _I, _I, _P]),
    "dfine_conv_fwd_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_conv1x1_accum_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_stream_fork": (c_int, [_P, _P]),
    "dfine_stream_create": (c_int, [c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "dfine_stream_destroy": (c_int, [_P]),
    "dfine_upload": (c_int, [_P, _P, _L, _P]),
    "dfine_conv_epilogue_supported": (c_int, [_I, _I, _I, _I, _I, _I]),
    "dfine_conv_affine_once": (c_int, [_P, _P, _P, _I]),
    "dfine_dwconv_affine_once": (c_int, [_P, _P, _P, _I]),
    "dfine_dwconv_affine_supported": (c_int, [_I, _I, _I, _I, _I, _I]),
    "dfine_conv_affine_supported": (c_int, [_I, _I, _I, _I, _I, _I]),
    "dfine_bn_fold": (c_int, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "dfine_conv_accum_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_conv_wgrad_ws_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "dfine_conv_wgrad_bf16": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_fdr_fwd": (c_int, [_P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_fdr_bwd": (c_int, [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_topk_anchors": (c_int, [_P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_conv1x1_seg_fwd_bf16": (c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_conv1x1_seg_accum_bf16": (c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_conv1x1_seg_accum_parts_bf16": (c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, ctypes.c_uint, _I, _I, _I, _I, _I, _P]),
    "dfine_conv1x1_seg_wgrad_bf16": (c_int, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_conv_wgrad_splits": (c_int, [_I, _I, _I, _I, _I, _I]),
    "dfine_linear_wgrad_splits": (c_int, [_I, _I, _I]),
    "dfine_multi_wgrad_reduce": (c_int, [_P, _I, _I, _P]),
    "dfine_multi_wgrad_reduce_blocks": (c_int, [_I, c_int64]),
    "dfine_linear_wgrad_group_row": (c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "dfine_linear_wgrad_group": (c_int, [_P, _I, _I, _P]),
    "dfine_conv_wgrad1_group_splits": (c_int, [_I, _I, _I, _I]),
    "dfine_conv_wgrad1_group_ws_floats": (_L, [_I, _I, _I, _I]),
    "dfine_conv_wgrad1_group_row": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_conv_wgrad1_group": (c_int, [_P, _I, _I, _P]),
    "dfine_linear_act_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_linear_dgrad_relu": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_multi_cast_bf16_t": (c_int, [_P, _I, _P]),
    "dfine_act_fwd_bf16": (c_int, [_P, _P, _L, _I, _P]),
    "dfine_act_bwd_bf16": (c_int, [_P, _P, _P, _L, _I, _P]),
    "dfine_attn_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "dfine_attn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                               _F, _P]),
    "dfine_attn_mask_summary_bytes": (_L, [_I]),
    "dfine_attn_mask_summary": (c_int, [_P, _I, _P, _P]),
    "dfine_attn_fwd_ms": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "dfine_attn_bwd_ms": (c_int, [_P] * 13 + [_I] * 12 + [_F, _P]),
    "dfine_attn_mask_bits_words": (_L, [_I]),
    "dfine_attn_mask_bits": (c_int, [_P, _I, _P, _P]),
    "dfine_groupnorm_ws_floats": (_L, [_I, _I, _I]),
    "dfine_groupnorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "dfine_groupnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_bilinear_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_bilinear_bwd": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_mask_loss_sums": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_mask_loss_grad": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_mask_cost": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "dfine_conv1x1_bw_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfine_conv_f32_packed_elems": (_L, [_I, _I, _I, _I]),
    "dfine_conv_f32_pack_weights": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "dfine_conv_f32_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_conv_f32_wgrad_splits": (c_int, [_I, _I, _I, _I, _I, _I]),
    "dfine_conv_f32_wgrad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_upsample2_zero_f32": (c_int, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "dfine_gemm_f32_nt": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, c_int64, c_int64, c_int64, _I, _I, _F, _I, _P]),
    "dfine_gemm_f32": (c_int, [_I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, c_int64, c_int64, c_int64, _I, _I, _F, _I, _P]),
    "dfine_gemm_f32_nn": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, c_int64, c_int64, c_int64, _F, _I, _P]),
    "dfine_colsum_f32_splits": (c_int, [_I]),
    "dfine_colsum_f32": (c_int, [_P, _P, _P, _P, _I, _I, _P]),
    "dfine_mask_bits_words": (c_int64, [c_int64]),
    "dfine_mask_pack_bits": (c_int, [_P, _I, _F, _I, c_int64, _P, _P]),
    "dfine_mask_iou_bits": (c_int, [_P, _P, _I, _I, c_int64, _P, _P]),
    "dfine_mosaic_place_u8": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_warp_affine_u8": (c_int, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "dfine_affine_boxes": (c_int, [_P, _P, _P, _I, _P, _F, _F, _F, _F, _P]),
    "dfine_preprocess_u8": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_postprocess": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_cdn_group": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "dfine_linear_wgrad_ws_floats": (_L, [_I, _I, _I]),
    "dfine_linear_wgrad_bf16": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dfine_ln_fused_fwd": (c_int, [_I, _P, _I, _P, _I, _P, _I, _P, _P, _F, _F, _P, _P, _P, _P, _L, _I, _P]),
    "dfine_ln_fused_bwd": (c_int, [_I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "dfine_ln_fused_bwd_ws_floats": (_L, [_L, _I]),
    "dfine_stem_supported": (c_int, [_I, _I, _I, _I]),
    "dfine_stem_pack_weights": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "dfine_stem_conv_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_dgrad_s2_bf16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_wgrad_ws_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "dfine_stem_wgrad_bf16": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_conv2_bf16": (c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_dgrad_s2_2_bf16": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_wgrad2_bf16": (c_int, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dfine_stem_pool_fwd": (c_int, [_P, _P, _L, _I, _I, _P]),
    "dfine_stem_pool_bwd": (c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "dfine_stem_pool_bwd_acc": (c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "dfine_bn_act_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
}
for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = library/header mismatch
    _fn.restype, _fn.argtypes = _res, _args

if _lib.dfine_abi_version() != ABI_VERSION:
    raise ImportError(f"libdfine_hip.so ABI {_lib.dfine_abi_version()} != binding ABI {ABI_VERSION}")

EXPORTED = tuple(_SIGNATURES)


class _PureQueries:
    """Memoized front of the library's size / split-count / capability queries (pure functions of a few integers that the
    launch wrappers ask for on every call: ~400 ctypes round trips per train step)."""

    def __getattr__(self, name):
        fn, cache = getattr(_lib, name), {}

        def call(*args):
            r = cache.get(args)
            if r is None:
                r = cache[args] = fn(*args)
            return r

        setattr(self, name, call)
        return call


_PURE = _PureQueries()

# ---- optional per-kernel timing (bench.py roofline leg): HIP events recorded on the launch
# stream (torch's current stream) right around the launch of the named entry points, together with the
# algorithmic work (FLOPs or bytes) of the launch.
_TIMED = {}          # name -> list of (start_event, end_event, work, bound_seconds)
MFMA_PEAK_FLOPS, HBM_PEAK_BYTES = 2.5e15, 8.0e12       # MI355X_MICROARCH.md: dense bf16 MFMA, HBM3E
F32_MFMA_PEAK_FLOPS = 157.3e12                           # f32-input MFMA (= the fp32 vector rate): the fp32 kernels' roof
_TIMING_ON = False   # bench.py switches this per step (`timing_active`): the steps it samples after its timed region
_TIMING_ISOLATED = False   # True: the sampled step runs everything on ONE stream (no weight gradients beside the chain) and its
                           # records go under "iso:<key>"; False: the step keeps its two streams and side-stream launches are
                           # bracketed by events on the side stream - the mode the un-instrumented steps run in


_FORCE_EAGER = False  # TrainStep runs backbone + encoder eagerly instead of replaying their graphs (bench.py's eager samples)


def force_eager(flag):
    global _FORCE_EAGER
    _FORCE_EAGER = bool(flag)


def enable_timing(names=None):
    """names: iterable of keys to record, or None = every instrumented launch."""
    global _TIMING_ON
    _TIMED.clear()
    _TIMED["*"] = None if names is None else set(names)
    _TIMING_ON = True


def timing_active(flag, isolated=False):
    global _TIMING_ON, _TIMING_ISOLATED
    _TIMING_ON = bool(flag) and "*" in _TIMED
    _TIMING_ISOLATED = bool(isolated) and _TIMING_ON


def disable_timing():
    global _TIMING_ON, _TIMING_ISOLATED
    _TIMED.clear()
    _TIMING_ON = _TIMING_ISOLATED = False


def timing_summary():
    """name -> (launches, mean_ms, total_ms, total_work, bound_ms) after a device synchronize; bound_ms = sum over the launches
    of max(FLOPs / MFMA peak, compulsory bytes / HBM peak) - the time a launch cannot go below on this chip."""
    torch.cuda.synchronize()
    out = {}
    for n, ev in _TIMED.items():
        if n == "*":
            continue
        tot = sum(e[0].elapsed_time(e[1]) for e in ev)
        out[n] = (len(ev), tot / max(len(ev), 1), tot, sum(e[2] for e in ev), sum(e[3] for e in ev) * 1e3)
    return out


class _timed:
    """`with _timed(key, work):` - no-op unless bench.py enabled timing for `key` and the current step is sampled."""
    __slots__ = ("ev", "a", "work", "bound", "stream")

    def __init__(self, name, work=0.0, io=0.0, stream=None, peak=None):
        """work: algorithmic FLOPs (bytes for the HBM-bound keys); io: compulsory HBM bytes of an MFMA-keyed launch;
        stream: the torch stream the launch goes to when that is not the current one (side-stream weight gradients);
        peak: matrix rate of the launch's arithmetic type when that is not bf16 (the fp32 kernels)."""
        self.ev = None
        if _TIMING_ON:
            want = _TIMED.get("*")
            if want is None or name in want:
                self.ev = _TIMED.setdefault("iso:" + name if _TIMING_ISOLATED else name, [])
                self.work = work
                self.bound = max(work / (peak or MFMA_PEAK_FLOPS), io / HBM_PEAK_BYTES) if io else 0.0
                self.stream = stream

    def __enter__(self):
        if self.ev is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(self.stream) if self.stream is not None else self.a.record()

    def __exit__(self, *exc):
        if self.ev is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(self.stream) if self.stream is not None else b.record()
            self.ev.append((self.a, b, self.work, self.bound))


timed = _timed
_DTYPE = {torch.float32: 0, torch.bfloat16: 1}


def _check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}: {_lib.dfine_last_error().decode()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw hipStream_t of torch's current stream (the private getters are ~20x cheaper than building a
    torch.cuda.Stream object per launch; ~1500 launches per step go through here)."""
    if _raw_stream is not None:
        return _raw_stream(_raw_device() if _raw_device is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class CaptureArena:
    """Pinned host memory a capturing segment owns: staging for the table uploads recorded in its graphs (dfine_upload)."""

    def __init__(self, nbytes=8 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.off = 0

    def upload(self, t, device):
        """CPU tensor -> device tensor through a slice of the arena, on the current stream."""
        t = t.contiguous()
        n = t.numel() * t.element_size()
        off = (self.off + 63) // 64 * 64
        if off + n > self.buf.numel():
            raise RuntimeError("capture arena exhausted (table uploads recorded in a HIP graph)")
        self.off = off + n
        host = self.buf[off:off + n].view(t.dtype).view(t.shape)
        host.copy_(t)
        dst = torch.empty(t.shape, dtype=t.dtype, device=device)
        _check(_lib.dfine_upload(dst.data_ptr(), host.data_ptr(), n, _stream()), "dfine_upload")
        return dst


def _ptr(t):
    """Device address as a plain int (ctypes converts it for the c_void_p parameters); None -> NULL."""
    return t.data_ptr() if t is not None else None


def _levels(shapes, points):
    hw = (c_int * (2 * len(shapes)))(*[v for s in shapes for v in s])
    pts = (c_int * len(points))(*points)
    return hw, pts


def _dtype_code(t):
    if t.dtype not in _DTYPE:
        raise TypeError(f"HIP kernels take float32 or bfloat16 tensors, got {t.dtype}")
    return _DTYPE[t.dtype]


# ------------------------------------------------------------------------------------- MSDA
def msda_forward(value, loc, weight, shapes, points):
    B, L, H, D = value.shape
    Lq = loc.shape[1]
    out = torch.empty(B, Lq, H * D, device=value.device, dtype=value.dtype)
    hw, pts = _levels(shapes, points)
    _check(_lib.dfine_msda_fwd(_ptr(value), _ptr(loc), _ptr(weight), _ptr(out), _dtype_code(value),
                               B, L, H, D, Lq, len(shapes), hw, pts, _stream()), "dfine_msda_fwd")
    return out


def _finish_grad_value(gv32, dtype):
    if dtype == torch.float32:
        return gv32
    out = torch.empty(gv32.shape, device=gv32.device, dtype=torch.bfloat16)
    _check(_lib.dfine_cast_f32_to_bf16(_ptr(gv32), _ptr(out), gv32.numel(), _stream()),
           "dfine_cast_f32_to_bf16")
    return out


def msda_backward(value, loc, weight, grad_out, shapes, points):
    B, L, H, D = value.shape
    Lq = loc.shape[1]
    gv = torch.zeros(B, L, H, D, device=value.device, dtype=torch.float32)
    gl = torch.empty_like(loc)
    gw = torch.empty_like(weight)
    hw, pts = _levels(shapes, points)
    if grad_out.dtype != value.dtype:
        grad_out = grad_out.to(value.dtype)
    _check(_lib.dfine_msda_bwd(_ptr(value), _ptr(loc), _ptr(weight), _ptr(grad_out), _ptr(gv),
                               _ptr(gl), _ptr(gw), _dtype_code(value), B, L, H, D, Lq,
                               len(shapes), hw, pts, _stream()), "dfine_msda_bwd")
    return _finish_grad_value(gv, value.dtype), gl, gw


def msda_algorithmic_bytes(batch, lq, heads=8, head_dim=32, points=12, elt=2, backward=False):
    """per image and layer, forward = gathered value reads Lq*H*P*4 corners*hd*elt + offsets
    Lq*H*P*2*elt + logits Lq*H*P*elt + reference boxes Lq*16 + output Lq*H*hd*elt; backward = the same gathered reads
    again + an equal volume of grad_value read-modify-write (3 x the gathered bytes) + the small tensors twice."""
    gathered = lq * heads * points * 4 * head_dim * elt
    small = lq * heads * points * 2 * elt + lq * heads * points * elt + lq * 16 + lq * heads * head_dim * elt
    return batch * ((3 * gathered + 2 * small) if backward else (gathered + small))


def msda_fused_forward(value, ref, offsets, logits, shapes, points, offset_scale):
    B, L, H, D = value.shape
    Lq = ref.shape[1]
    if offsets.dtype != value.dtype:
        offsets = offsets.to(value.dtype)
    if logits.dtype != value.dtype:
        logits = logits.to(value.dtype)
    out = torch.empty(B, Lq, H * D, device=value.device, dtype=value.dtype)
    hw, pts = _levels(shapes, points)
    with _timed("msda_fwd", msda_algorithmic_bytes(B, Lq, H, D, sum(points), value.element_size())):
        _check(_lib.dfine_msda_fused_fwd(_ptr(value), _ptr(ref), _ptr(offsets), _ptr(logits), _ptr(out),
                                         _dtype_code(value), B, L, H, D, Lq, len(shapes), hw, pts,
                                         float(offset_scale), _stream()), "dfine_msda_fused_fwd")
    return out


# How d(value) of the deformable attention is accumulated: 2 = scaled f16, one packed atomic per channel pair (default: half
# the atomic dwords of f32), 3 = int32 fixed point in 64-bit integer atomics (exact, order-independent), 0 = f32 atomics.
# Default (-1): 2 for a bf16 model (the result is rounded to bf16 anyway), 0 for fp32 math (the fixed-point form resolves
# ~1e-5 of max |grad_out| under its worst-case overflow bound: fine for training, coarser than the f32 atomics' 1e-7).
MSDA_ACC_MODE = -1       # (tests/test_msda_gpu.py, tests/test_bf16_anchor_gpu.py and tools/msda_acc_bench.py set it)


def msda_grad_value_buffer(value, uses=1):
    """Zero-filled accumulator for d(value) (the backward kernels add into it with atomics); `uses` = backward calls that
    will share it.  Modes 2 / 3: a flat f16 / int32 tensor with 16 trailing bytes of scale state."""
    n = value.numel()
    mode = MSDA_ACC_MODE if MSDA_ACC_MODE >= 0 else (2 if value.dtype == torch.bfloat16 else 0)
    if mode in (2, 3) and n % 2 == 0:
        buf = (torch.zeros(n + 8, device=value.device, dtype=torch.float16) if mode == 2
               else torch.zeros(n + 4, device=value.device, dtype=torch.int32))
        buf._dfine_fx = (tuple(value.shape), max(int(uses), 1), n)
        return buf
    return torch.zeros(value.shape, device=value.device, dtype=torch.float32)


_ACC_MODE_OF = {torch.float32: 0, torch.float16: 2, torch.int32: 3}


def msda_finish_grad_value(acc, dtype):
    mode = _ACC_MODE_OF[acc.dtype]
    if mode == 0:
        return _finish_grad_value(acc, dtype)
    shape, _, n = acc._dfine_fx
    out = torch.empty(shape, device=acc.device, dtype=dtype)
    _check(_lib.dfine_cast_scaled_acc(_ptr(acc), _ptr(out), mode, _DTYPE[dtype], n, acc.data_ptr() + acc.element_size() * n,
                                      _stream()), "dfine_cast_scaled_acc")
    return out


def msda_fused_backward(value, ref, offsets, logits, grad_out, shapes, points, offset_scale, gv_acc=None):
    """gv_acc: a shared fp32 accumulator (msda_grad_value_buffer) several backward calls on the SAME value add into; then
    the first return value is None and the caller finishes the buffer once (msda_finish_grad_value)."""
    B, L, H, D = value.shape
    Lq = ref.shape[1]
    off_dtype, log_dtype = offsets.dtype, logits.dtype
    if offsets.dtype != value.dtype:
        offsets = offsets.to(value.dtype)
    if logits.dtype != value.dtype:
        logits = logits.to(value.dtype)
    if grad_out.dtype != value.dtype:
        grad_out = grad_out.to(value.dtype)
    gv = msda_grad_value_buffer(value) if gv_acc is None else gv_acc
    mode = _ACC_MODE_OF[gv.dtype]
    fx_state, hit_bound = None, 1.0
    if mode != 0:
        fx_state = gv.data_ptr() + gv.element_size() * gv._dfine_fx[2]
        hit_bound = float(gv._dfine_fx[1] * Lq)
    goff = torch.empty_like(offsets)
    glog = torch.empty_like(logits)
    hw, pts = _levels(shapes, points)
    with _timed("msda_bwd", msda_algorithmic_bytes(B, Lq, H, D, sum(points), value.element_size(), backward=True)):
        _check(_lib.dfine_msda_fused_bwd_acc(_ptr(value), _ptr(ref), _ptr(offsets), _ptr(logits),
                                             _ptr(grad_out), _ptr(gv), _ptr(goff), _ptr(glog),
                                             _dtype_code(value), B, L, H, D, Lq, len(shapes), hw, pts,
                                             float(offset_scale), mode, fx_state, hit_bound, _stream()), "dfine_msda_fused_bwd_acc")
    return (None if gv_acc is not None else msda_finish_grad_value(gv, value.dtype)), goff.to(off_dtype), glog.to(log_dtype)


# ------------------------------------------------------------------------------------- denoising group (A4)
def cdn_group(labels, boxes, offsets, flip_rand, rnd_cls, sign01, mag, bs, gmax, groups, num_classes, flip_below, box_noise_scale):
    """labels int64 [T], boxes f32 [T, 4] (the batch's targets concatenated), offsets int32 [bs + 1] on the device, the four random
    tensors of the reference's draw order -> (cls int32 [bs, total], box_unact f32 [bs, total, 4]), total = 2 * groups * gmax
    (dfine_cdn_group: one launch, bit-identical to the torch composition)."""
    total = 2 * groups * gmax
    dev = offsets.device
    assert labels.dtype == torch.int64 and boxes.dtype == torch.float32 and offsets.dtype == torch.int32
    assert flip_rand.dtype == torch.float32 and rnd_cls.dtype == torch.int32 and sign01.dtype == torch.float32 and mag.dtype == torch.float32
    for t, shape in ((flip_rand, (bs, total)), (rnd_cls, (bs, total)), (sign01, (bs, total, 4)), (mag, (bs, total, 4))):
        assert tuple(t.shape) == shape and t.is_contiguous()
    cls = torch.empty(bs, total, device=dev, dtype=torch.int32)
    unact = torch.empty(bs, total, 4, device=dev, dtype=torch.float32)
    _check(_lib.dfine_cdn_group(_ptr(labels.contiguous()), _ptr(boxes.contiguous()), _ptr(offsets), _ptr(flip_rand), _ptr(rnd_cls),
                                _ptr(sign01), _ptr(mag), _ptr(cls), _ptr(unact), bs, gmax, groups, int(num_classes), float(flip_below),
                                float(box_noise_scale), _stream()), "dfine_cdn_group")
    return cls, unact


# ------------------------------------------------------------------------------------- matcher
def hungarian_assign(logits, boxes, tgt_labels, tgt_boxes, sizes, w_class, w_bbox, w_giou, alpha,
                     gamma, use_focal=True, extra_cost=None):
    if not use_focal:
        raise NotImplementedError("the HIP matcher implements the focal class cost (reference default)")
    K, B, Q, C = logits.shape
    dev = logits.device
    T = int(sum(sizes))
    tmax = int(max(sizes)) if sizes else 0
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + int(n))
    cols = torch.full((K, T), -1, device=dev, dtype=torch.int32)
    if T == 0:
        return cols, None
    tgt_offset = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    logits = logits.float().contiguous()
    boxes = boxes.float().contiguous()
    tgt_labels = tgt_labels.to(torch.int64).contiguous()
    tgt_boxes = tgt_boxes.float().contiguous()
    cost = torch.empty(K, B, tmax, Q, device=dev, dtype=torch.float32)
    extra = None
    if extra_cost is not None:  # caller layout [K,B,Q,Tmax] -> target-major
        extra = extra_cost.float().permute(0, 1, 3, 2).contiguous()
    _check(_lib.dfine_match(_ptr(logits), _ptr(boxes), _ptr(tgt_labels), _ptr(tgt_boxes),
                            _ptr(tgt_offset), _ptr(extra), _ptr(cost), c_void_p(0), _ptr(cols),
                            K, B, Q, C, tmax, T, float(w_class), float(w_bbox), float(w_giou),
                            float(alpha), float(gamma), _stream()), "dfine_match")
    cols._dfine_tgt_offset = tgt_offset          # (the criterion's device-side plan builder reads the same offsets)
    return cols, cost.permute(0, 1, 3, 2)


def lsap(cost_tq, sizes):
    """cost_tq [K, B, Tmax, Q] f32 (target-major) -> cols int32 [K, T]."""
    K, B, tmax, Q = cost_tq.shape
    T = int(sum(sizes))
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + int(n))
    cols = torch.full((K, T), -1, device=cost_tq.device, dtype=torch.int32)
    if T == 0:
        return cols
    tgt_offset = torch.tensor(offs, dtype=torch.int32).to(cost_tq.device)
    cost_tq = cost_tq.float().contiguous()
    _check(_lib.dfine_lsap(_ptr(cost_tq), _ptr(tgt_offset), c_void_p(0), _ptr(cols), K, B, Q, tmax,
                           T, _stream()), "dfine_lsap")
    return cols


# ------------------------------------------------------------------------------------- depthwise conv
def dwconv_forward(x, w, stride, pad):
    B, C, H, W = x.shape
    K = w.shape[-1]
    OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y = torch.empty(B, C, OH, OW, device=x.device, dtype=x.dtype)
    with _timed("dfine_dwconv_fwd"):
        _check(_lib.dfine_dwconv_fwd(_ptr(x), _ptr(w), _ptr(y), _dtype_code(x), B, C, H, W, K, stride, pad,
                                     _stream()), "dfine_dwconv_fwd")
    return y


def dwconv_affine_supported(x, K, stride, pad):
    return x.dim() == 4 and bool(_lib.dfine_dwconv_affine_supported(_dtype_code(x), x.shape[2], x.shape[3], K, stride, pad))


def dwconv_forward_affine(x, w, stride, pad, scale, shift, act, lab=None):
    """lab[0] * act(scale[c] * dwconv(x)[c] + shift[c]) + lab[1] before the store (dfine_dwconv_affine_once + dfine_dwconv_fwd):
    depthwise conv -> eval-mode BatchNorm -> activation as one launch.  Shapes of dwconv_affine_supported only."""
    _check(_lib.dfine_dwconv_affine_once(scale.data_ptr(), shift.data_ptr(), _ptr(lab), _ACT_CODE[act]), "dfine_dwconv_affine_once")
    try:
        return _DW_FWD_IMPL(x, w, stride, pad)
    finally:
        _lib.dfine_dwconv_affine_once(None, None, None, 0)


_DW_FWD_IMPL = dwconv_forward               # (the module attribute is a shim while a program is exported)


def _dw_zeros(shape, dev):
    """Zero-filled fp32 [C, 1, K, K] for a depthwise weight gradient (the kernel adds into it).  In a captured backward the ~20
    depthwise layers of a step take slices of ONE zero-filled arena: one fill launch on the main stream instead of one each."""
    if CAPTURE_DUAL is None:
        return torch.zeros(shape, device=dev, dtype=torch.float32)
    n = 1
    for d in shape:
        n *= int(d)
    n_al = (n + 63) // 64 * 64
    ar = getattr(CAPTURE_DUAL, "dw_arena", None)
    if ar is None or ar[1] + n_al > ar[0].numel():
        ar = [torch.zeros(max(1 << 18, n_al), device=dev, dtype=torch.float32), 0]
        CAPTURE_DUAL.dw_arena = ar
    off = ar[1]
    ar[1] = off + n_al
    return ar[0][off:off + n].view(shape)


def dwconv_acc_supported(x, K, stride, pad):
    B, C, H, W = x.shape
    return x.dtype == torch.bfloat16 and K == 3 and stride == 2 and pad == 1 and H % 2 == 0 and W % 8 == 0 and W <= 320


def dwconv_backward(x, w, dy, stride, pad, need_dx=True, need_dw=True, side_dw=False, acc=None):
    """side_dw: the weight gradient is launched on the side stream (see _side_fork); the caller guarantees that its only
    consumer runs behind side_join() (the fused optimizer's gradient gather).
    acc: the gradient of x from its other consumer (dwconv_acc_supported shapes): the data gradient is added onto it in place."""
    B, C, H, W = x.shape
    K = w.shape[-1]
    if acc is not None and need_dx:
        _check(_lib.dfine_dwconv_s2_dgrad_acc(_ptr(w), _ptr(dy), _ptr(acc), B, C, H, W, _stream()), "dfine_dwconv_s2_dgrad_acc")
        _, dw = dwconv_backward(x, w, dy, stride, pad, False, need_dw, side_dw) if need_dw else (None, None)
        return acc, dw
    dx = torch.empty_like(x) if need_dx else None
    dw = _dw_zeros(w.shape, x.device) if need_dw else None
    if side_dw and need_dw and _side_ok("dw"):
        st = _side_fork(x.device, direct=True)
        _check(_lib.dfine_dwconv_bwd(_ptr(x), _ptr(w), _ptr(dy), None, _ptr(dw), _dtype_code(x), B, C, H, W, K, stride, pad,
                                     st.cuda_stream), "dfine_dwconv_bwd")
        _SIDE_LIVE.append((x, w, dy))         # (not dw: autograd only takes ownership of a gradient nobody else references - it would CLONE it, on the main stream)
        if need_dx:
            _check(_lib.dfine_dwconv_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), None, _dtype_code(x), B, C, H, W, K, stride, pad,
                                         _stream()), "dfine_dwconv_bwd")
        return dx, dw
    with _timed("dfine_dwconv_bwd"):
        _check(_lib.dfine_dwconv_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _dtype_code(x), B, C, H, W,
                                     K, stride, pad, _stream()), "dfine_dwconv_bwd")
    return dx, dw


# ------------------------------------------------------------------------------------- fused BN
_ACT = {None: 0, "relu": 1, "silu": 2, "swish": 2}
_BN_WS = {}


def _bn_workspace(dev, nfloats):
    """Scratch for the partial sums: consumed inside the same call by stream-ordered kernels, so one
    growing buffer per (device, stream) is reused by every BatchNorm unit instead of an allocation each."""
    key = (dev.index, _stream())
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() < nfloats:
        ws = torch.empty(max(nfloats, 1 << 16), device=dev, dtype=torch.float32)
        _BN_WS[key] = ws
    return ws


_BN_WS_NEED = {}


def _bn_ws_need(B, C, HW):
    key = (B, C, HW)
    n = _BN_WS_NEED.get(key)
    if n is None:
        n = _BN_WS_NEED[key] = int(_lib.dfine_bn_ws_floats(B, C, HW))
    return n


def bn_residual_supported(x):
    return x.dtype == torch.bfloat16 and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 8 == 0 and x.shape[1] <= 4096 and x.numel() // 8 < (1 << 31)


def bn_act_forward(x, gamma, beta, running_mean, running_var, lab_scale, lab_bias, act, training,
                   momentum, eps, residual=None):
    """x [B, C, H, W] contiguous.  Returns (y, saved) where saved feeds bn_act_backward.
    (133 calls per train step: pointers of the four `stats` rows are computed, not sliced, and the HIP-event timing
    wrapper is skipped unless a bench asked for it.)"""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    dev = x.device
    y = torch.empty_like(x)
    stats = torch.empty(4, C, device=dev, dtype=torch.float32)     # mean, invstd, scale, shift
    ws = _bn_workspace(dev, _bn_ws_need(B, C, HW))
    sp = stats.data_ptr()
    row = 4 * C
    if residual is not None:       # y = unit(x) + residual in the apply pass (bn_residual_supported shapes; one-shot request)
        _lib.dfine_bn_residual_once(residual.data_ptr())
    status = _lib.dfine_bn_act_fwd(x.data_ptr(), y.data_ptr(), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                   _ptr(running_var), _ptr(lab_scale), _ptr(lab_bias), sp, sp + row, sp + 2 * row,
                                   sp + 3 * row, ws.data_ptr(), _DTYPE[x.dtype], B, C, HW, _ACT[act],
                                   1 if training else 0, float(momentum), float(eps), _stream())
    if status != 0:
        _check(status, "dfine_bn_act_fwd")
    return y, stats                # (eval mode: rows 0 / 1 hold the running mean and rsqrt(running_var + eps), written by the kernel)


_EPI_OK = {}


def conv_epilogue_supported(B, cin, cout, H, W, ks):
    """Does the forward kernel for this shape have the accumulate epilogue (dfine_conv_epilogue_supported)?"""
    key = (B, cin, cout, H, W, ks)
    ok = _EPI_OK.get(key)
    if ok is None:
        ok = _EPI_OK[key] = bool(_lib.dfine_conv_epilogue_supported(B, cin, cout, H, W, ks))
    return ok


def conv_accumulate_bf16(x, w2, y, ks):
    """y += conv(x) (bf16, in place): dfine_conv_accum_bf16, shapes of conv_epilogue_supported only."""
    B, cin, H, W = x.shape
    status = _lib.dfine_conv_accum_bf16(x.data_ptr(), w2.data_ptr(), y.data_ptr(), B, cin, y.shape[1], H, W, ks, _stream())
    if status != 0:
        _check(status, "dfine_conv_accum_bf16")
    return y


def bn_act_backward(x, dy, stats, lab_scale, act, training, need_affine=True, need_lab=True, dlab_ptr=None):
    """-> (dx, dgamma, dbeta, dlab).  dgamma / dbeta are two separate tensors (autograd's AccumulateGrad takes ownership
    of a whole tensor but has to copy a view: 2 x 133 small device copies per D-FINE-m step).  `dlab_ptr`: device address
    of two adjacent floats the kernel ADDS the learnable-affine gradients to (the fused optimizer's flat gradient slots);
    then no dlab tensor is made."""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    dev = x.device
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32) if need_affine else None
    dbeta = torch.empty(C, device=dev, dtype=torch.float32) if need_affine else None
    dlab = torch.zeros(2, device=dev, dtype=torch.float32) if (need_lab and dlab_ptr is None) else None
    ws = _bn_workspace(dev, _bn_ws_need(B, C, HW))
    sp = stats.data_ptr()
    row = 4 * C
    dl = dlab_ptr if (need_lab and dlab_ptr is not None) else _ptr(dlab)
    status = _lib.dfine_bn_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), sp, sp + row, sp + 2 * row, sp + 3 * row,
                                   _ptr(lab_scale), _ptr(dgamma), _ptr(dbeta), dl, ws.data_ptr(),
                                   _DTYPE[x.dtype], B, C, HW, _ACT[act], 1 if training else 0, _stream())
    if status != 0:
        _check(status, "dfine_bn_act_bwd")
    return dx, dgamma, dbeta, dlab


def bn2_supported(x):
    """Shape / dtype of a conv output the RepVGG-unit kernels take (bf16 NCHW, H*W % 8 == 0, moderate channel count)."""
    if x.dtype != torch.bfloat16 or x.dim() != 4:
        return False
    B, C = x.shape[0], x.shape[1]
    return bool(_PURE.dfine_bn2_supported(B, C, x.numel() // max(B * C, 1)))


def bn2_act_forward(x1, x2, residual, bn1, bn2, act):
    """act(BN_1(x1) + BN_2(x2)) [+ residual] with batch statistics; bn = (gamma, beta, running_mean, running_var,
    momentum, eps).  -> (y, saved [8, C])."""
    B, C = x1.shape[0], x1.shape[1]
    HW = x1.numel() // max(B * C, 1)
    y = torch.empty_like(x1)
    saved = torch.empty(8, C, device=x1.device, dtype=torch.float32)
    ws = _bn_workspace(x1.device, int(_PURE.dfine_bn2_ws_floats(B, C, HW)))
    _check(_lib.dfine_bn2_act_fwd(_ptr(x1), _ptr(x2), _ptr(residual), _ptr(y), _ptr(bn1[0]), _ptr(bn1[1]), _ptr(bn1[2]),
                                  _ptr(bn1[3]), _ptr(bn2[0]), _ptr(bn2[1]), _ptr(bn2[2]), _ptr(bn2[3]), _ptr(saved), _ptr(ws),
                                  B, C, HW, _ACT[act], float(bn1[4]), float(bn1[5]), float(bn2[4]), float(bn2[5]), _stream()),
           "dfine_bn2_act_fwd")
    return y, saved


def bn2_act_backward(x1, x2, dy, saved, act, need_affine=(True, True)):
    B, C = x1.shape[0], x1.shape[1]
    HW = x1.numel() // max(B * C, 1)
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    dev = x1.device
    g = [torch.empty(C, device=dev, dtype=torch.float32) if need_affine[i // 2] else None for i in range(4)]
    ws = _bn_workspace(dev, int(_PURE.dfine_bn2_ws_floats(B, C, HW)))
    _check(_lib.dfine_bn2_act_bwd(_ptr(x1), _ptr(x2), _ptr(dy), _ptr(dx1), _ptr(dx2), _ptr(saved), _ptr(g[0]), _ptr(g[1]),
                                  _ptr(g[2]), _ptr(g[3]), _ptr(ws), B, C, HW, _ACT[act], _stream()), "dfine_bn2_act_bwd")
    return dx1, dx2, g[0], g[1], g[2], g[3]


# ------------------------------------------------------------------------------------- losses
def _view3(t):
    """(ptr, batch stride, query stride) of a [B, Q, inner] view with unit inner stride."""
    if t is None:
        return c_void_p(0), 0, 0
    assert t.dim() == 3 and t.stride(2) == 1, "loss kernels need a unit inner stride"
    return _ptr(t), t.stride(0), t.stride(1)


def head_losses(logits, boxes, corners, ref, teacher_corners, teacher_logits, cls_plan, box_plan,
                tgt_labels, tgt_boxes, wtable, reg_max, reg_scale, alpha, gamma, temp, s_vfl, s_l1,
                s_giou, s_fgl, c_pos, c_neg, scales_dev=None, box_count_dev=None, zbytes=None):
    """One launch group for all losses of a head.  Returns (out[5], grad_logits, grad_l1, grad_giou,
    grad_corners_fgl, grad_corners_ddf) - see dfine_head_losses in include/dfine_hip.h.
    zbytes: a ZERO-FILLED uint8 block of head_losses_zbytes(...) bytes (16-byte aligned) to use for the packed outputs - the
    caller cleared the blocks of all heads with one fill (dfine_head_losses_prezeroed_once)."""
    B, Q, C = logits.shape
    dev = logits.device
    dt = logits.dtype
    m_cls, m_box = int(cls_plan.shape[1]), int(box_plan.shape[1])
    # [out(8 f32) | grad_l1 | grad_giou | map_cls | map_box | pad to 16 B | grad_corners_fgl]: everything the call wants
    # zeroed, back to back in one allocation - the library clears it with ONE fill (see dfine_head_losses)
    nq = B * Q
    maps_end = 32 + nq * 40
    fgl_off = (maps_end + 15) // 16 * 16
    nb = corners.shape[-1] if corners is not None else 0
    if zbytes is None:
        zbytes = torch.empty(fgl_off + nq * nb * dt.itemsize, device=dev, dtype=torch.uint8)
    else:
        assert zbytes.dtype == torch.uint8 and zbytes.numel() >= fgl_off + nq * nb * dt.itemsize and zbytes.data_ptr() % 16 == 0
        zbytes = zbytes[:fgl_off + nq * nb * dt.itemsize]
        _lib.dfine_head_losses_prezeroed_once()
    zbuf = zbytes[:32 + nq * 32].view(torch.float32)
    out = zbuf[:5]
    g_box = zbuf[8:].view(2, B, Q, 4)
    scratch_i = zbytes[32 + nq * 32: maps_end].view(torch.int32).view(2, nq)
    g_logits = torch.empty(B, Q, C, device=dev, dtype=dt)
    scratch_f = torch.empty(m_cls + m_box + B * Q, device=dev, dtype=torch.float32)
    g_fgl = g_ddf = None
    if corners is not None:
        g_fgl = zbytes[fgl_off:].view(dt).view(B, Q, nb)
        if teacher_corners is not None:
            g_ddf = torch.empty(B, Q, nb, device=dev, dtype=dt)
    lp, lsb, lsq = _view3(logits)
    bp, bsb, bsq = _view3(boxes)
    cp, csb, csq = _view3(corners)
    rp, rsb, rsq = _view3(ref)
    tcp, tcsb, tcsq = _view3(teacher_corners)
    tlp, tlsb, tlsq = _view3(teacher_logits)
    wt = (c_float * len(wtable))(*wtable) if wtable is not None else None
    if scales_dev is not None:
        # the six scalar factors (and the length of the box plan) stay on the device: no host value depends on the matching
        _check(_lib.dfine_head_losses_dev(
            lp, lsb, lsq, bp, bsb, bsq, cp, csb, csq, rp, rsb, rsq, tcp, tcsb, tcsq, tlp, tlsb, tlsq,
            _ptr(cls_plan), m_cls, _ptr(box_plan), m_box, _ptr(tgt_labels), _ptr(tgt_boxes), wt,
            int(reg_max), float(reg_scale), float(alpha), float(gamma), float(temp), _ptr(scales_dev), _ptr(box_count_dev),
            _ptr(g_logits), _ptr(g_box[0]), _ptr(g_box[1]), _ptr(g_fgl), _ptr(g_ddf), _ptr(scratch_f[:m_cls]),
            _ptr(scratch_f[m_cls:m_cls + m_box]), _ptr(scratch_i[0]), _ptr(scratch_i[1]),
            _ptr(scratch_f[m_cls + m_box:]), _ptr(out), _dtype_code(logits), B, Q, C, _stream()),
            "dfine_head_losses_dev")
        return out, g_logits, g_box[0], g_box[1], g_fgl, g_ddf
    _check(_lib.dfine_head_losses(
        lp, lsb, lsq, bp, bsb, bsq, cp, csb, csq, rp, rsb, rsq, tcp, tcsb, tcsq, tlp, tlsb, tlsq,
        _ptr(cls_plan), m_cls, _ptr(box_plan), m_box, _ptr(tgt_labels), _ptr(tgt_boxes), wt,
        int(reg_max), float(reg_scale), float(alpha), float(gamma), float(temp), float(s_vfl),
        float(s_l1), float(s_giou), float(s_fgl), float(c_pos), float(c_neg), _ptr(g_logits),
        _ptr(g_box[0]), _ptr(g_box[1]), _ptr(g_fgl), _ptr(g_ddf), _ptr(scratch_f[:m_cls]),
        _ptr(scratch_f[m_cls:m_cls + m_box]), _ptr(scratch_i[0]), _ptr(scratch_i[1]),
        _ptr(scratch_f[m_cls + m_box:]), _ptr(out), _dtype_code(logits), B, Q, C, _stream()),
        "dfine_head_losses")
    return out, g_logits, g_box[0], g_box[1], g_fgl, g_ddf


def head_losses_zbytes(B, Q, corner_bins, itemsize):
    """Bytes of the packed zero-initialised output block of one head_losses call, rounded up to a multiple of 256."""
    nq = B * Q
    n = (32 + nq * 40 + 15) // 16 * 16 + nq * corner_bins * itemsize
    return (n + 255) // 256 * 256


def head_grads_scale(g, g_logits, g_l1, g_giou, g_fgl, g_ddf):
    """In-place backward scaling of the gradients head_losses left behind (dfine_head_grads_scale): g = d / d out[5]."""
    _check(_lib.dfine_head_grads_scale(
        _ptr(g), _ptr(g_logits), g_logits.numel(), _ptr(g_l1), _ptr(g_giou), g_l1.numel(), _ptr(g_fgl), _ptr(g_ddf),
        0 if g_fgl is None else g_fgl.numel(), _dtype_code(g_logits), _stream()), "dfine_head_grads_scale")


def criterion_plans_supported(K, tmax, Q):
    return bool(_PURE.dfine_criterion_plans_supported(int(K), int(tmax), int(Q)))


def criterion_plans(cols, tgt_offset, sizes, Q, want_float_count=False):
    """cols int32 [K, T] (device; every target matched), tgt_offset int32 [B + 1] (device), sizes = targets per image (host)
    -> (head_plans int64 [K, 3, T], go_plan int64 [3, K * T], go_count int32 [1], go_count_f float [1] or None): the
    criterion's gather plans and the GO union built on the device (csrc/plans.hip), no host synchronisation."""
    K, T = cols.shape
    B = len(sizes)
    dev = cols.device
    cap = K * T
    head_plans = torch.empty(K, 3, T, device=dev, dtype=torch.int64)
    go_plan = torch.empty(3, cap, device=dev, dtype=torch.int64)
    go_count = torch.empty(1, device=dev, dtype=torch.int32)
    go_f = torch.empty(1, device=dev, dtype=torch.float32) if want_float_count else None
    ws = torch.empty(int(_PURE.dfine_criterion_plans_ws_ints(K, T, B)), device=dev, dtype=torch.int32)
    _check(_lib.dfine_criterion_plans(_ptr(cols), _ptr(tgt_offset), K, T, B, int(Q), int(max(sizes)), _ptr(head_plans),
                                      _ptr(go_plan), cap, _ptr(go_count), _ptr(go_f), _ptr(ws), _stream()), "dfine_criterion_plans")
    return head_plans, go_plan, go_count, go_f


def criterion_scales(params, go_count, go_sum, world):
    """params: device float64 [R, 12] (see dfine_criterion_scales) -> scales float32 [R, 6] on the device."""
    R = params.shape[0]
    scales = torch.empty(R, 6, device=params.device, dtype=torch.float32)
    _check(_lib.dfine_criterion_scales(_ptr(params), R, _ptr(go_count), _ptr(go_sum), int(world), _ptr(scales), _stream()),
           "dfine_criterion_scales")
    return scales


# ------------------------------------------------------------------------------------- optimizer
def grad_sqnorm_buffer(device):
    """Result + scratch buffer of grad_sqnorm (element 0 is the squared norm)."""
    return torch.zeros(int(_lib.dfine_grad_sqnorm_ws_floats()), device=device, dtype=torch.float32)


def grad_sqnorm(flat_grad, grad_scale, out):
    _check(_lib.dfine_grad_sqnorm(_ptr(flat_grad), flat_grad.numel(), float(grad_scale), _ptr(out), _stream()),
           "dfine_grad_sqnorm")


def adamw_ema_step(param, grad, exp_avg, exp_avg_sq, ema, sqnorm, lr, beta1, beta2, eps, weight_decay, step,
                   grad_scale, max_norm, ema_momentum):
    _check(_lib.dfine_adamw_ema_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(ema),
                                     param.numel(), _ptr(sqnorm), float(lr), float(beta1), float(beta2), float(eps),
                                     float(weight_decay), int(step), float(grad_scale), float(max_norm),
                                     float(ema_momentum), _stream()), "dfine_adamw_ema_step")


def ema_update(ema, src, momentum):
    _check(_lib.dfine_ema_update(_ptr(ema), _ptr(src), ema.numel(), float(momentum), _stream()), "dfine_ema_update")


# ------------------------------------------------------------------------------------- MFMA conv
def conv_pack_weights(weight_f32, dgrad):
    cout, cin, ks, _ = weight_f32.shape
    n = int(_lib.dfine_conv_packed_elems(cout, cin, ks, 1 if dgrad else 0))
    w2 = torch.empty(n, device=weight_f32.device, dtype=torch.bfloat16)
    _check(_lib.dfine_conv_pack_weights(_ptr(weight_f32), _ptr(w2), cout, cin, ks, 1 if dgrad else 0, _stream()),
           "dfine_conv_pack_weights")
    return w2


def conv_packed_elems(cout, cin, ks, dgrad):
    return int(_lib.dfine_conv_packed_elems(cout, cin, ks, 1 if dgrad else 0))


def conv_pack_weights_multi(table, n_entries):
    """table: device int64 [n_entries, 8] (see dfine_conv_pack_weights_multi)."""
    _check(_lib.dfine_conv_pack_weights_multi(_ptr(table), n_entries, _stream()), "dfine_conv_pack_weights_multi")


def conv_forward_bf16(x, w2, cout, ks):
    """x [B, Cin, H, W] bf16 contiguous, w2 packed by conv_pack_weights -> y [B, cout, H, W] bf16."""
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.bfloat16)
    if not _TIMING_ON:                     # ~250 calls per step: no timing object, no FLOP arithmetic on the plain path
        status = _lib.dfine_conv_fwd_bf16(x.data_ptr(), w2.data_ptr(), y.data_ptr(), B, cin, cout, H, W, ks, _stream())
        if status != 0:
            _check(status, "dfine_conv_fwd_bf16")
        return y
    with _timed(f"conv{ks}x{ks}", 2.0 * B * H * W * cin * cout * ks * ks, io=2.0 * B * H * W * (cin + cout) + 2.0 * cin * cout * ks * ks):
        _check(_lib.dfine_conv_fwd_bf16(_ptr(x), _ptr(w2), _ptr(y), B, cin, cout, H, W, ks, _stream()),
               "dfine_conv_fwd_bf16")
    return y


_AFF_OK = {}
_ACT_CODE = {None: 0, "relu": 1, "silu": 2, "swish": 2}
_CONV_FWD_IMPL = conv_forward_bf16          # (the module attribute is a shim while a program is exported)


def conv_affine_supported(B, cin, cout, H, W, ks):
    """Does the forward kernel for this shape take the affine + activation epilogue (dfine_conv_affine_supported)?"""
    key = (B, cin, cout, H, W, ks)
    ok = _AFF_OK.get(key)
    if ok is None:
        ok = _AFF_OK[key] = bool(_lib.dfine_conv_affine_supported(B, cin, cout, H, W, ks))
    return ok


def bn_fold(gamma, beta, running_mean, running_var, eps):
    """-> (scale, shift) fp32 [C] of an eval-mode BatchNorm: gamma / sqrt(var + eps), beta - mean * scale (one launch)."""
    C = running_mean.numel()
    scale = torch.empty(C, device=running_mean.device, dtype=torch.float32)
    shift = torch.empty(C, device=running_mean.device, dtype=torch.float32)
    _check(_lib.dfine_bn_fold(_ptr(gamma), _ptr(beta), running_mean.data_ptr(), running_var.data_ptr(), float(eps), C, scale.data_ptr(),
                              shift.data_ptr(), _stream()), "dfine_bn_fold")
    return scale, shift


def conv_forward_affine(x, w2, cout, ks, scale, shift, act, lab=None):
    """lab[0] * act(scale[n] * conv(x)[n] + shift[n]) + lab[1] in the convolution's store phase (dfine_conv_affine_once +
    dfine_conv_fwd_bf16): conv -> eval-mode BatchNorm / deployed bias -> activation as ONE launch.  Shapes of
    conv_affine_supported only.  scale / shift fp32 [cout], lab fp32 [2] or None, act in (None, "relu", "silu")."""
    _check(_lib.dfine_conv_affine_once(scale.data_ptr(), shift.data_ptr(), _ptr(lab), _ACT_CODE[act]), "dfine_conv_affine_once")
    try:
        return _CONV_FWD_IMPL(x, w2, cout, ks)
    finally:
        _lib.dfine_conv_affine_once(None, None, None, 0)          # (consumed by the launch; withdrawn if it never happened)


def conv1x1_seg_forward_affine(x_parts, w2, cout, scale, shift, act, lab=None):
    """The same epilogue behind the 1x1 convolution over the channel concatenation of `x_parts` (read in place) -> new [B, cout, H, W]."""
    B, _, H, W = x_parts[0].shape
    y = torch.empty(B, cout, H, W, device=x_parts[0].device, dtype=torch.bfloat16)
    _check(_lib.dfine_conv_affine_once(scale.data_ptr(), shift.data_ptr(), _ptr(lab), _ACT_CODE[act]), "dfine_conv_affine_once")
    try:
        _SEG_FORWARD_IMPL(tuple(x_parts), w2, (y,))
    finally:
        _lib.dfine_conv_affine_once(None, None, None, 0)
    return y


def maps_to_tokens(maps):
    """List of [B, C, H_l, W_l] bf16 contiguous maps -> tokens [B, sum(H_l W_l), C] (levels in list order)."""
    B, C = maps[0].shape[0], maps[0].shape[1]
    hws = [m.shape[2] * m.shape[3] for m in maps]
    L = sum(hws)
    tokens = torch.empty(B, L, C, device=maps[0].device, dtype=torch.bfloat16)
    row = 0
    for m, hw in zip(maps, hws):
        _check(_lib.dfine_maps_tokens_bf16(_ptr(m), _ptr(tokens), B, C, hw, L, row, 1, _stream()), "dfine_maps_tokens_bf16")
        row += hw
    return tokens


def embedding_backward(g, idx, rows, padding_idx=-1, stream=None):
    """g [.., D] f32, idx [..] int32 / int64 -> dw [rows, D] f32 = gradient of F.embedding(idx, weight) (small tables: dfine_embedding_bwd)."""
    D = g.shape[-1]
    g = g.reshape(-1, D)
    if g.dtype != torch.float32 or not g.is_contiguous():
        g = g.float().contiguous()
    idx = idx.reshape(-1).contiguous()
    dw = torch.empty(rows, D, device=g.device, dtype=torch.float32)
    _check(_lib.dfine_embedding_bwd(_ptr(g), _ptr(idx), 32 if idx.dtype == torch.int32 else 64, _ptr(dw), g.shape[0], rows, D, int(padding_idx),
                                    _stream() if stream is None else stream), "dfine_embedding_bwd")
    return dw


def upsample2_nearest(x, backward=False):
    """x [B, C, H, W] bf16 contiguous -> [B, C, 2H, 2W] (nearest); backward=True: x is the [B, C, 2H, 2W] gradient, returns
    the [B, C, H, W] sums of its 2 x 2 blocks."""
    B, C, H, W = x.shape
    if backward:
        out = torch.empty(B, C, H // 2, W // 2, device=x.device, dtype=x.dtype)
        _check(_lib.dfine_upsample2_nearest_bf16(_ptr(out), _ptr(x), B * C, H // 2, W // 2, 1, _stream()), "dfine_upsample2_nearest_bf16")
        return out
    out = torch.empty(B, C, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
    _check(_lib.dfine_upsample2_nearest_bf16(_ptr(x), _ptr(out), B * C, H, W, 0, _stream()), "dfine_upsample2_nearest_bf16")
    return out


def tokens_to_maps(tokens, shapes, outs=None):
    """tokens [B, L, C] bf16 contiguous -> one contiguous [B, C, h, w] map per (h, w) in `shapes`; `outs`: optional
    preallocated maps (entries may be None) to write into."""
    B, L, C = tokens.shape
    out, row = [], 0
    for i, (h, w) in enumerate(shapes):
        m = outs[i] if outs is not None else None
        if m is None or m.shape != (B, C, h, w) or m.dtype != torch.bfloat16 or not m.is_contiguous():
            m = torch.empty(B, C, h, w, device=tokens.device, dtype=torch.bfloat16)
        _check(_lib.dfine_maps_tokens_bf16(_ptr(m), _ptr(tokens), B, C, h * w, L, row, 0, _stream()), "dfine_maps_tokens_bf16")
        out.append(m)
        row += h * w
    return out


def conv1x1_accumulate(x, w2, y):
    """y += conv1x1(x) (bf16, packed weights); False when the shape needs the separate-add fallback."""
    B, cin, H, W = x.shape
    with _timed("conv1x1", 2.0 * B * H * W * cin * y.shape[1], io=2.0 * B * H * W * (cin + y.shape[1]) + 2.0 * cin * y.shape[1]):
        status = _lib.dfine_conv1x1_accum_bf16(_ptr(x), _ptr(w2), _ptr(y), B, cin, y.shape[1], H * W, _stream())
    if status == -1:
        return False
    _check(status, "dfine_conv1x1_accum_bf16")
    return True


def _seg_arrays(parts):
    hw = parts[0].shape[2] * parts[0].shape[3]
    ptrs = (c_void_p * len(parts))(*[t.data_ptr() for t in parts])
    chans = (c_int * len(parts))(*[t.shape[1] for t in parts])
    bstr = (c_int * len(parts))(*[(t.stride(0) // hw if t.shape[0] > 1 else t.shape[1]) for t in parts])
    return ptrs, chans, bstr


def is_channel_part(t):
    """[B, C, H, W] bf16 that is contiguous or a channel slice of a contiguous tensor (planes dense, images equally spaced)."""
    if t.dim() != 4 or t.dtype != torch.bfloat16:
        return False
    B, C, H, W = t.shape
    hw = H * W
    return (t.stride(3) == 1 or W == 1) and (t.stride(2) == W or H == 1) and (t.stride(1) == hw or C == 1) and (
        B == 1 or (t.stride(0) % hw == 0 and t.stride(0) >= C * hw)) and t.data_ptr() % 16 == 0


def conv1x1_seg_forward(x_parts, w2, y_parts, accum=False):
    """1x1 conv over the channel concatenation of `x_parts` ([B, C_k, H, W] bf16, contiguous or channel slices) written into
    the channel concatenation `y_parts` (preallocated; accum: added onto it), packed weights `w2` ([sum C_out] x [sum C_in])."""
    B, _, H, W = x_parts[0].shape
    cin, cout = sum(t.shape[1] for t in x_parts), sum(t.shape[1] for t in y_parts)
    xp, xc, xb = _seg_arrays(x_parts)
    yp, yc, yb = _seg_arrays(y_parts)
    if accum not in (False, True):          # a sequence of flags, one per output part
        mask = sum(1 << k for k, a in enumerate(accum) if a)
        extra = sum(t.shape[1] for t, a in zip(y_parts, accum) if a)
        with _timed("conv1x1", 2.0 * B * H * W * cin * cout, io=2.0 * B * H * W * (cin + cout + extra) + 2.0 * cin * cout):
            _check(_lib.dfine_conv1x1_seg_accum_parts_bf16(xp, xc, xb, len(x_parts), _ptr(w2), yp, yc, yb, len(y_parts), mask, B, cin,
                                                           cout, H, W, _stream()), "dfine_conv1x1_seg_accum_parts_bf16")
        return
    fn, name = ((_lib.dfine_conv1x1_seg_accum_bf16, "dfine_conv1x1_seg_accum_bf16") if accum
                else (_lib.dfine_conv1x1_seg_fwd_bf16, "dfine_conv1x1_seg_fwd_bf16"))
    with _timed("conv1x1", 2.0 * B * H * W * cin * cout, io=2.0 * B * H * W * (cin + (2 if accum else 1) * cout) + 2.0 * cin * cout):
        _check(fn(xp, xc, xb, len(x_parts), _ptr(w2), yp, yc, yb, len(y_parts), B, cin, cout, H, W, _stream()), name)


_SEG_FORWARD_IMPL = conv1x1_seg_forward      # (the module attribute conv1x1_seg_forward is a shim while a program is exported)


def conv1x1_seg_wgrad(x_parts, dy, partials=False):
    B, _, H, W = x_parts[0].shape
    cin, cout = sum(t.shape[1] for t in x_parts), dy.shape[1]
    dw = None if partials else torch.empty(cout, cin, 1, 1, device=dy.device, dtype=torch.float32)
    ws = torch.empty(int(_PURE.dfine_conv_wgrad_ws_floats(B, cin, cout, H, W, 1)), device=dy.device, dtype=torch.float32)
    xp, xc, xb = _seg_arrays(x_parts)
    if partials and _side_ok("seg"):
        st = _side_fork(dy.device)
        with _timed("conv1x1_wgrad", 2.0 * B * H * W * cin * cout, io=2.0 * B * H * W * (cin + cout) + 4.0 * cin * cout, stream=st.stream):
            _check(_lib.dfine_conv1x1_seg_wgrad_bf16(xp, xc, xb, len(x_parts), _ptr(dy), None, _ptr(ws), B, cin, cout, H, W,
                                                     st.cuda_stream), "dfine_conv1x1_seg_wgrad_bf16")
        _SIDE_LIVE.append((x_parts, dy, ws))
        return ws, (int(_PURE.dfine_conv_wgrad_splits(B, cin, cout, H, W, 1)), cout, cin, 1, _p16(cout), _p16(cin))
    with _timed("conv1x1_wgrad", 2.0 * B * H * W * cin * cout, io=2.0 * B * H * W * (cin + cout) + 4.0 * cin * cout):
        _check(_lib.dfine_conv1x1_seg_wgrad_bf16(xp, xc, xb, len(x_parts), _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W,
                                                 _stream()), "dfine_conv1x1_seg_wgrad_bf16")
    if partials:
        return ws, (int(_PURE.dfine_conv_wgrad_splits(B, cin, cout, H, W, 1)), cout, cin, 1, _p16(cout), _p16(cin))
    return dw


# ---- weight gradients on a second HIP stream ---------------------------------------------------------------------------
# In backward a convolution's weight gradient depends on nothing after it and nothing depends on it until the optimizer's
# flush, while the chain  BatchNorm backward -> data gradient -> next unit's BatchNorm backward ...  is a sequence of short,
# latency-bound launches that leave most of the chip idle.  The deferred weight-gradient launches (partial sums, reduced at
# the flush) therefore go to a side stream forked from the current one and joined at the flush: they fill the idle CUs
# under the chain.  Inputs stay referenced until the join (the caching allocator only knows the stream a block was
# allocated on).  DFINE_WGRAD_STREAM=0: everything on the current stream.
WGRAD_STREAM = os.environ.get("DFINE_WGRAD_STREAM", "1") == "1"
_SIDE = {}
_SIDE_LIVE = []
_SIDE_PRIORITY = 0        # stream priority of the side stream (torch: lower number = higher priority; measured: no effect - tools/probe/main_prio.py)
_SIDE_GROUP_AT = 32       # registered 1x1 convolution problems that trigger an early grouped launch (12 / 48 / never: +0.13 / 0 / +0.4 ms
                          # per step; tools/ab_step.py with AB_RECAPTURE=1 - the value is baked into the captured backward graphs)
_SIDE_LINEAR_GROUP_AT = 128   # the same for the token-stream linears of the (eager) decoder backward: 64 until round 6; with the 128-tile
                          # grouped kernel, in-process A/B against 64: 96 -0.05, 128 -0.08 / -0.11 / -0.19, 192 -0.11, never -0.07 ms per step


CAPTURE_SIDE = False      # set by dl.engine.GraphedSegment while it captures: the side stream is forked into the capture (and joined
                          # before the capture ends), so the weight-gradient launches keep their second stream inside the graph


CAPTURE_DUAL = None       # set by dl.engine.GraphedSegment while it captures a backward pass as a CHAIN of graph pairs: the side
                          # stream records its own graphs next to the main stream's (no event between two captures); the object's
                          # side_launch() is told about every side-stream launch and decides where one pair ends and the next begins


def _side_ok(kind=None):
    """kind: dw (depthwise), seg (part-wise 1x1), conv1, conv3, group (grouped 1x1), linear (grouped token-stream linears), reduce
    (split reductions), stem - every kind of weight gradient pays its way on the side stream (measured: all 29.1 ms per step,
    none 29.7, any subset 29.6-30.5), so the kind no longer selects anything."""
    return (WGRAD_STREAM and not _TIMING_ISOLATED
            and (CAPTURE_SIDE or CAPTURE_DUAL is not None or not torch.cuda.is_current_stream_capturing()))


def side_stream_ok():
    return _side_ok()


class _SideStream:
    """The side stream of one device: the torch object (for `with torch.cuda.stream(...)`) and its raw handle."""
    __slots__ = ("stream", "cuda_stream")

    def __init__(self, dev):
        if _SIDE_PRIORITY > 0:
            # lower than the default stream: torch.cuda.Stream has no such level, the library creates it (kept for the process)
            raw = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _check(_lib.dfine_stream_create(_SIDE_PRIORITY, ctypes.byref(raw)), "dfine_stream_create")
            self.stream = torch.cuda.ExternalStream(raw.value, device=dev)
        else:
            self.stream = torch.cuda.Stream(device=dev, priority=_SIDE_PRIORITY)
        self.cuda_stream = self.stream.cuda_stream


def _side_fork(dev, direct=False):
    """-> the side stream (`.cuda_stream` = raw handle), made to wait for everything enqueued on the current stream so far.
    direct: the launch writes a finished gradient tensor (depthwise, stem), not partial sums for the deferred split reduction."""
    st = _SIDE.get(dev.index)
    if st is None:
        st = _SIDE[dev.index] = _SideStream(dev)
    if CAPTURE_DUAL is not None:
        CAPTURE_DUAL.side_launch(direct)   # the replay orders the pair: main graph, event, side graph (dl/engine.py)
        return st
    if _lib.dfine_stream_fork(_stream(), st.cuda_stream) != 0:
        _check(-2, "dfine_stream_fork")
    return st


def side_stream(dev):
    """The side stream object of a device (made on first use), without any ordering."""
    st = _SIDE.get(dev.index)
    if st is None:
        st = _SIDE[dev.index] = _SideStream(dev)
    return st


def stream_wait(src, dst):
    """Raw stream `dst` waits for everything enqueued on raw stream `src` so far."""
    if _lib.dfine_stream_fork(src, dst) != 0:
        _check(-2, "dfine_stream_fork")


def side_join():
    """The current stream waits for the side stream's launches (called before their results are consumed)."""
    if CAPTURE_DUAL is not None:
        return                           # the replay joins once, after the last pair; the inputs stay referenced until then
    if _SIDE_LIVE:
        cur = _stream()
        for st in _SIDE.values():
            if _lib.dfine_stream_fork(st.cuda_stream, cur) != 0:
                _check(-2, "dfine_stream_fork")
        _SIDE_LIVE.clear()


def conv_wgrad_supported(H, W, ks):
    """3x3: rows are walked in 16-byte chunks, so the kernel wants W % 8 == 0; narrower maps (the 20x20 level) are run on
    zero-padded copies of x and dy (conv_wgrad_bf16): zero columns of dy add nothing and zero columns of x are the
    convolution's own padding.  1x1: the pixels of a plane are walked in 16-byte chunks; planes with H * W % 8 != 0 (the
    10x10 level of a 320 x 320 input) run on copies padded to the next multiple of 8 pixels."""
    return (ks == 3 and W % 2 == 0 and (W + 7) // 8 * 8 <= 160) or ks == 1


def _p16(n):
    return (n + 15) // 16 * 16


def conv_wgrad_bf16(x, dy, ks, partials=False):
    """x [B,Cin,H,W], dy [B,Cout,H,W] bf16 contiguous -> dw [Cout,Cin,ks,ks] f32; partials=True: (ws, meta) for a deferred
    dfine_multi_wgrad_reduce, meta = (splits, Cout, Cin, taps, NP16, CP16)."""
    st = None
    need_pad = ks == 3 and x.shape[3] % 8
    if ks == 1 and (x.shape[2] * x.shape[3]) % 8:
        # a 1x1 weight gradient is a sum over pixels: the planes as rows of hw pixels, zero-padded below like the 3x3 case
        hw = x.shape[2] * x.shape[3]
        x, dy = x.reshape(x.shape[0], x.shape[1], 1, hw), dy.reshape(dy.shape[0], dy.shape[1], 1, hw)
        need_pad = True
    if need_pad:
        padw = (x.shape[3] + 7) // 8 * 8 - x.shape[3]
        if partials and _side_ok("conv3" if ks == 3 else "conv1"):
            # the zero-padded copies are part of the weight-gradient work: made on the side stream too (their blocks belong to
            # its allocator pool and stay referenced until the join)
            st = _side_fork(x.device)
            with torch.cuda.stream(st.stream):
                xp, dyp = torch.nn.functional.pad(x, (0, padw)), torch.nn.functional.pad(dy, (0, padw))
            _SIDE_LIVE.append((x, dy))
            x, dy = xp, dyp
        else:
            x = torch.nn.functional.pad(x, (0, padw))
            dy = torch.nn.functional.pad(dy, (0, padw))
    B, cin, H, W = x.shape
    cout = dy.shape[1]
    dw = None if partials else torch.empty(cout, cin, ks, ks, device=x.device, dtype=torch.float32)
    if partials and ks == 1 and (H * W) % 8 == 0:
        # registered only: every 1x1 weight gradient of a flush runs in one launch (linear_wgrad_flush -> dfine_conv_wgrad1_group);
        # a problem of a grouped launch is cut into fewer splits than a stand-alone one (dfine_conv_wgrad1_group_splits)
        ws = torch.empty(int(_PURE.dfine_conv_wgrad1_group_ws_floats(B, cin, cout, H * W)), device=x.device, dtype=torch.float32)
        _CW_PENDING.append((x, dy, ws, B, cin, cout, H * W))
        if len(_CW_PENDING) >= _SIDE_GROUP_AT and _side_ok("group"):
            _flush_conv_group(True)          # ... or in a few, on the side stream while backward goes on
        return ws, (int(_PURE.dfine_conv_wgrad1_group_splits(B, cin, cout, H * W)), cout, cin, 1, _p16(cout), _p16(cin))
    ws = torch.empty(int(_PURE.dfine_conv_wgrad_ws_floats(B, cin, cout, H, W, ks)), device=x.device, dtype=torch.float32)
    if partials and _side_ok("conv3" if ks == 3 else "conv1"):
        if st is None:
            st = _side_fork(x.device)
        with _timed(f"conv{ks}x{ks}_wgrad", 2.0 * B * H * W * cin * cout * ks * ks,
                    io=2.0 * B * H * W * (cin + cout) + 4.0 * cin * cout * ks * ks, stream=st.stream):
            _check(_lib.dfine_conv_wgrad_bf16(_ptr(x), _ptr(dy), None, _ptr(ws), B, cin, cout, H, W, ks, st.cuda_stream), "dfine_conv_wgrad_bf16")
        _SIDE_LIVE.append((x, dy, ws))
        return ws, (int(_PURE.dfine_conv_wgrad_splits(B, cin, cout, H, W, ks)), cout, cin, ks * ks, _p16(cout), _p16(cin))
    with _timed(f"conv{ks}x{ks}_wgrad", 2.0 * B * H * W * cin * cout * ks * ks, io=2.0 * B * H * W * (cin + cout) + 4.0 * cin * cout * ks * ks):
        _check(_lib.dfine_conv_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W, ks, _stream()),
               "dfine_conv_wgrad_bf16")
    if partials:
        return ws, (int(_PURE.dfine_conv_wgrad_splits(B, cin, cout, H, W, ks)), cout, cin, ks * ks, _p16(cout), _p16(cin))
    return dw


def multi_wgrad_reduce_blocks(splits, elems):
    return int(_PURE.dfine_multi_wgrad_reduce_blocks(int(splits), int(elems)))


def multi_wgrad_reduce(table, n_entries, max_blocks, io=0.0, side=False):
    """io: bytes of partial sums + destinations the launch streams (its roofline is HBM; no FLOPs of its own).
    side: on the side stream, behind the weight-gradient launches already queued there."""
    if side and _side_ok("reduce"):
        st = _side_fork(table.device)             # (forked after the table's upload was enqueued on the current stream)
        with _timed("wgrad_reduce", 0.0, io=io, stream=st.stream):
            _check(_lib.dfine_multi_wgrad_reduce(_ptr(table), n_entries, int(max_blocks), st.cuda_stream), "dfine_multi_wgrad_reduce")
        _SIDE_LIVE.append((table,))
        return
    with _timed("wgrad_reduce", 0.0, io=io):
        _check(_lib.dfine_multi_wgrad_reduce(_ptr(table), n_entries, int(max_blocks), _stream()), "dfine_multi_wgrad_reduce")


# ------------------------------------------------------------------------------------- FDR head
def fdr_forward(corners, ref, wtable, reg_scale, k=4):
    n = corners.numel() // corners.shape[-1]
    reg_max = corners.shape[-1] // 4 - 1
    dev = corners.device
    boxes = torch.empty(n, 4, device=dev, dtype=torch.float32)
    stat = torch.empty(n, 4 * (k + 1), device=dev, dtype=torch.float32)
    idx = torch.empty(n * 4 * k, device=dev, dtype=torch.uint8)
    wt = (c_float * len(wtable))(*wtable)
    _check(_lib.dfine_fdr_fwd(_ptr(corners), _ptr(ref), wt, float(reg_scale), _ptr(boxes), _ptr(stat), _ptr(idx),
                              _dtype_code(corners), n, reg_max, k, _stream()), "dfine_fdr_fwd")
    return boxes, stat, idx


def fdr_backward(corners, ref, wtable, reg_scale, g_boxes, g_stat, idx, k=4):
    n = corners.numel() // corners.shape[-1]
    reg_max = corners.shape[-1] // 4 - 1
    g = torch.empty_like(corners)
    wt = (c_float * len(wtable))(*wtable)
    _check(_lib.dfine_fdr_bwd(_ptr(corners), _ptr(ref), wt, float(reg_scale), _ptr(g_boxes), _ptr(g_stat), _ptr(idx),
                              _ptr(g), _dtype_code(corners), n, reg_max, k, _stream()), "dfine_fdr_bwd")
    return g


def multi_copy_f32(srcs, dst_offsets, dst_flat, chunk=1 << 16, add=False):
    """Copies (add: adds) the fp32 tensors `srcs` to dst_flat[dst_offsets[i] : +numel] with one launch."""
    import numpy as np
    rows = []
    for t, off in zip(srcs, dst_offsets):
        n, p = t.numel(), t.data_ptr()
        for c0 in range(0, n, chunk):
            rows.append((p + 4 * c0, off + c0, min(chunk, n - c0)))
    if not rows:
        return
    from .d_fine.arch.utils import upload
    table = upload(np.asarray(rows, dtype=np.int64), dst_flat.device)
    fn = _lib.dfine_multi_add_f32 if add else _lib.dfine_multi_copy_f32
    _check(fn(_ptr(table), len(rows), _ptr(dst_flat), _stream()), "dfine_multi_add_f32" if add else "dfine_multi_copy_f32")
    return table      # keep alive until the stream has consumed it


def sum_f32(tensors):
    """Sum of 2 .. 8 contiguous fp32 tensors of one shape (numel % 4 == 0) in one pass -> a new tensor."""
    n = len(tensors)
    out = torch.empty_like(tensors[0])
    ptrs = (c_void_p * n)(*[t.data_ptr() for t in tensors])
    _check(_lib.dfine_sum_f32(ptrs, n, out.data_ptr(), out.numel(), _stream()), "dfine_sum_f32")
    return out


def multi_cast_bf16(table, n_entries):
    """table: device int64 [n_entries, 3] = (fp32 src pointer, bf16 dst pointer, element count)."""
    _check(_lib.dfine_multi_cast_bf16(_ptr(table), n_entries, _stream()), "dfine_multi_cast_bf16")


# ------------------------------------------------------------------------------------- query selection
def topk_anchors(logits, k, with_scores=False):
    """logits [B, Q, C] (unit class stride) -> indices [B, k] i64 of the k anchors with the largest
    max-over-classes logit, descending (ties: ascending index)."""
    assert logits.dim() == 3 and logits.stride(2) == 1
    B, Q, C = logits.shape
    idx = torch.empty(B, k, device=logits.device, dtype=torch.int64)
    sc = torch.empty(B, k, device=logits.device, dtype=torch.float32) if with_scores else None
    _check(_lib.dfine_topk_anchors(_ptr(logits), logits.stride(0), logits.stride(1), _ptr(idx), _ptr(sc),
                                   _dtype_code(logits), B, Q, C, k, _stream()), "dfine_topk_anchors")
    return (idx, sc) if with_scores else idx


def preprocess_u8(frames, out_hw, resized_hw, top_left=(0, 0), pad_value=114, dtype=torch.float32):
    """frames uint8 [B, Hs, Ws, 3] BGR (device) -> [B, 3, Ho, Wo] `dtype`, RGB / 255 (resize to `resized_hw`, placed at `top_left`)."""
    B, Hs, Ws, _ = frames.shape
    out = torch.empty(B, 3, out_hw[0], out_hw[1], device=frames.device, dtype=dtype)
    _check(_lib.dfine_preprocess_u8(_ptr(frames), _ptr(out), _DTYPE[dtype], B, Hs, Ws, out_hw[0], out_hw[1], resized_hw[0],
                                    resized_hw[1], top_left[0], top_left[1], int(pad_value), _stream()), "dfine_preprocess_u8")
    return out


def postprocess(logits, boxes, k, height, width, to_round=True):
    """logits [B, Q, C] f32/bf16, boxes [B, Q, 4] f32 cxcywh (normalised) -> labels [B, k] i64, query [B, k] i64,
    boxes [B, k, 4] f32 absolute xyxy, scores [B, k] f32 - the k best (query, class) pairs per image, descending."""
    B, Q, C = logits.shape
    logits = logits.contiguous()
    boxes = boxes.float().contiguous()
    dev = logits.device
    labels = torch.empty(B, k, device=dev, dtype=torch.int64)
    query = torch.empty(B, k, device=dev, dtype=torch.int64)
    out_boxes = torch.empty(B, k, 4, device=dev, dtype=torch.float32)
    scores = torch.empty(B, k, device=dev, dtype=torch.float32)
    _check(_lib.dfine_postprocess(_ptr(logits), _ptr(boxes), _ptr(labels), _ptr(query), _ptr(out_boxes), _ptr(scores),
                                  _dtype_code(logits), B, Q, C, k, int(height), int(width), int(bool(to_round)),
                                  _stream()), "dfine_postprocess")
    return labels, query, out_boxes, scores


# ------------------------------------------------------------------------------------- token-stream linears / attention
ACT_CODE = {None: 0, "none": 0, "relu": 1, "gelu": 2, "silu": 3}


def linear_act(x2d, w, bias=None, act=0, out_f32=False, out=None):
    """x2d [M, K] bf16 (unit inner stride), w [N, K] bf16 (unit inner stride; row stride may exceed K: slices and
    transposed shadows), bias fp32 [N] or None -> act(x2d @ w.T + bias) [M, N] bf16 (fp32 when out_f32)."""
    M, K = x2d.shape
    N = w.shape[0]
    assert w.shape[1] == K and (K == 1 or (x2d.stride(1) == 1 and w.stride(1) == 1))     # size-1 dims carry arbitrary strides
    assert x2d.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    ldx = x2d.stride(0) if M > 1 else K
    ldw = w.stride(0) if N > 1 else K
    if out is None:
        out = torch.empty(M, N, device=x2d.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    if not _TIMING_ON:                     # ~160 calls per step
        status = _lib.dfine_linear_act_fwd(x2d.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, ldx, ldw,
                                           out.stride(0) if M > 1 else N, int(act), int(out.dtype == torch.float32), _stream())
        if status != 0:
            _check(status, "dfine_linear_act_fwd")
        return out
    with _timed("linear", 2.0 * M * N * K, io=2.0 * (M * K + N * K + M * N)):
        _check(_lib.dfine_linear_act_fwd(_ptr(x2d), _ptr(w), _ptr(bias), _ptr(out), M, N, K, ldx, ldw,
                                         out.stride(0) if M > 1 else N, int(act), int(out.dtype == torch.float32), _stream()),
               "dfine_linear_act_fwd")
    return out


def linear_dgrad_relu(d2, w_t, relu_out):
    """(d2 [M, K] @ w_t [N, K]^T) masked by relu_out [M, N] > 0 -> [M, N] bf16: the data gradient of the layer behind a
    Linear + ReLU with the ReLU's backward in the epilogue (dfine_linear_dgrad_relu)."""
    M, K = d2.shape
    N = w_t.shape[0]
    assert w_t.shape[1] == K and (K == 1 or (d2.stride(1) == 1 and w_t.stride(1) == 1)) and relu_out.shape == (M, N) and relu_out.is_contiguous()
    assert d2.dtype == torch.bfloat16 and w_t.dtype == torch.bfloat16 and relu_out.dtype == torch.bfloat16
    out = torch.empty(M, N, device=d2.device, dtype=torch.bfloat16)
    with _timed("linear", 2.0 * M * N * K, io=2.0 * (M * K + N * K + 2 * M * N)):
        _check(_lib.dfine_linear_dgrad_relu(_ptr(d2), _ptr(w_t), _ptr(relu_out), _ptr(out), M, N, K, d2.stride(0) if M > 1 else K,
                                            w_t.stride(0) if N > 1 else K, N, _stream()), "dfine_linear_dgrad_relu")
    return out


def multi_cast_bf16_t(table, n_entries):
    _check(_lib.dfine_multi_cast_bf16_t(_ptr(table), n_entries, _stream()), "dfine_multi_cast_bf16_t")


def act_forward(z, act):
    y = torch.empty_like(z)
    _check(_lib.dfine_act_fwd_bf16(_ptr(z), _ptr(y), z.numel(), int(act), _stream()), "dfine_act_fwd_bf16")
    return y


def act_backward(dy, ref, act):
    out = torch.empty_like(dy)
    _check(_lib.dfine_act_bwd_bf16(_ptr(dy), _ptr(ref), _ptr(out), dy.numel(), int(act), _stream()), "dfine_act_bwd_bf16")
    return out


def _ld(t):
    assert t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1), "need [B, L, C] rows with one stride"
    return t.stride(1)


def _attn_hd(hd):
    """Head dim the kernels run a head of `hd` channels on: 32 or 64 (zero-padded copies otherwise)."""
    return 32 if hd <= 32 else 64


def _pad_heads(t, num_heads, hd):
    """[B, L, H * hd] -> [B, L, H * P], P = 32 or 64, with every head zero-padded to P channels (head dims 8 / 16 / 24 run on the
    head-dim-32 kernels, 40 / 48 / 56 - the AIFI layer of D-FINE-x has 48 - on the head-dim-64 ones: zero channels add nothing to
    q k^T, and the extra output channels of v are dropped)."""
    B, L, _ = t.shape
    P = _attn_hd(hd)
    return torch.nn.functional.pad(t.reshape(B, L, num_heads, hd), (0, P - hd)).reshape(B, L, num_heads * P)


def attn_forward(q, k, v, num_heads, mask=None):
    """q, k, v [B, L, H * hd] bf16 views (e.g. column slices of the packed projection), hd = 32 (or 8 / 16 / 24: padded copies)
    -> o [B, L, H * hd] bf16, lse2 [B, H, L]."""
    B, L, E = q.shape
    hd = E // num_heads
    msum = None if mask is None else _mask_forms(mask)[1]
    if hd not in (32, 64):
        P = _attn_hd(hd)
        qp, kp, vp = (_pad_heads(t, num_heads, hd) for t in (q, k, v))
        o = torch.empty(B, L, num_heads * P, device=q.device, dtype=torch.bfloat16)
        lse2 = torch.empty(B, num_heads, L, device=q.device, dtype=torch.float32)
        _check(_lib.dfine_attn_fwd_ms(_ptr(qp), _ptr(kp), _ptr(vp), _ptr(o), _ptr(lse2), _ptr(mask), _ptr(msum), B, L, num_heads, P, _ld(qp),
                                      _ld(kp), _ld(vp), num_heads * P, float(hd) ** -0.5, _stream()), "dfine_attn_fwd_ms")
        return o.reshape(B, L, num_heads, P)[..., :hd].reshape(B, L, E), lse2
    o = torch.empty(B, L, E, device=q.device, dtype=torch.bfloat16)
    lse2 = torch.empty(B, num_heads, L, device=q.device, dtype=torch.float32)
    with _timed("attention", 4.0 * B * num_heads * L * L * hd, io=2.0 * 4 * B * num_heads * L * hd):
        _check(_lib.dfine_attn_fwd_ms(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse2), _ptr(mask), _ptr(msum), B, L, num_heads, hd, _ld(q),
                                      _ld(k), _ld(v), E, float(hd) ** -0.5, _stream()), "dfine_attn_fwd_ms")
    return o, lse2


_MASK_BITS = [None, None]          # (key of the last mask, its transposed bit mask): the decoder's layers share one mask per step


def _mask_bits(mask):
    """Transposed bit-packed form of a [L, L] uint8 mask for the dK / dV kernel (dfine_attn_mask_bits), remade when the mask
    tensor or its contents change."""
    return _mask_forms(mask)[0]


_MASK_SUMMARY = True       # tile summaries of the attention mask (tools/probe/attn_real_mask.py flips it for its A/B)


def _mask_forms(mask):
    """(transposed bit mask, tile summaries) of a [L, L] uint8 mask, remade when the mask tensor or its contents change: the
    decoder's layers share one mask per step.  The summaries (dfine_attn_mask_summary) let the kernels skip blocked tiles."""
    key = (mask.data_ptr(), mask._version, mask.shape[0], mask.device.index)
    if _MASK_BITS[0] != key:
        L = mask.shape[0]
        bits = torch.empty(int(_lib.dfine_attn_mask_bits_words(L)), device=mask.device, dtype=torch.int32)
        _check(_lib.dfine_attn_mask_bits(_ptr(mask), L, _ptr(bits), _stream()), "dfine_attn_mask_bits")
        msum = None
        if _MASK_SUMMARY:
            msum = torch.empty(int(_lib.dfine_attn_mask_summary_bytes(L)), device=mask.device, dtype=torch.uint8)
            _check(_lib.dfine_attn_mask_summary(_ptr(mask), L, _ptr(msum), _stream()), "dfine_attn_mask_summary")
        _MASK_BITS[0], _MASK_BITS[1] = key, (bits, msum, mask)       # (the mask is kept alive with its key)
    return _MASK_BITS[1][0], _MASK_BITS[1][1]


def attn_backward(q, k, v, o, dout, lse2, num_heads, dq, dk, dv, mask=None):
    """Writes dq, dk, dv ([B, L, H * 32] bf16 views, e.g. column slices of one packed gradient buffer)."""
    B, L, E = q.shape
    hd = E // num_heads
    delta = torch.empty_like(lse2)
    mbits, msum = (None, None) if mask is None else _mask_forms(mask)
    if hd not in (32, 64):
        P = _attn_hd(hd)
        qp, kp, vp, op, dop = (_pad_heads(t, num_heads, hd) for t in (q, k, v, o, dout))
        g = torch.empty(3, B, L, num_heads * P, device=q.device, dtype=torch.bfloat16)
        _check(_lib.dfine_attn_bwd_ms(_ptr(qp), _ptr(kp), _ptr(vp), _ptr(op), _ptr(dop), _ptr(lse2), _ptr(mask), _ptr(mbits), _ptr(msum), _ptr(g[0]),
                                      _ptr(g[1]), _ptr(g[2]), _ptr(delta), B, L, num_heads, P, _ld(qp), _ld(kp), _ld(vp), _ld(op), _ld(dop),
                                      _ld(g[0]), _ld(g[1]), _ld(g[2]), float(hd) ** -0.5, _stream()), "dfine_attn_bwd_ms")
        for dst, src in zip((dq, dk, dv), g):
            dst.copy_(src.reshape(B, L, num_heads, P)[..., :hd].reshape(B, L, E))
        return
    with _timed("attention", 10.0 * B * num_heads * L * L * hd, io=2.0 * 8 * B * num_heads * L * hd):
        _check(_lib.dfine_attn_bwd_ms(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(dout), _ptr(lse2), _ptr(mask), _ptr(mbits), _ptr(msum), _ptr(dq),
                                      _ptr(dk), _ptr(dv), _ptr(delta), B, L, num_heads, hd, _ld(q), _ld(k), _ld(v), _ld(o), _ld(dout),
                                      _ld(dq), _ld(dk), _ld(dv), float(hd) ** -0.5, _stream()), "dfine_attn_bwd_ms")


# ------------------------------------------------------------------------------------- linear wgrad
_LW_WS = {}


_CW_PENDING = []            # (x, dy, ws, B, Cin, Cout, HW): 1x1 convolution weight gradients registered since the last flush
_LW_PENDING = []            # (x2d, dy2d, ws, M, N, K) registered since the last linear_wgrad_flush


def linear_wgrad_partials(x2d, dy2d):
    """Partial sums only: (ws, weight meta, bias meta, bias offset in floats) for a deferred dfine_multi_wgrad_reduce.  The
    GEMM itself is only REGISTERED here (the tensors are kept alive); `linear_wgrad_flush` - called by the fused optimizer in
    front of every deferred reduction - runs all registered problems as one launch (dfine_linear_wgrad_group)."""
    M, K = x2d.shape
    N = dy2d.shape[1]
    ws = torch.empty(int(_PURE.dfine_linear_wgrad_ws_floats(M, N, K)), device=x2d.device, dtype=torch.float32)
    _LW_PENDING.append((x2d, dy2d, ws, M, N, K))
    if len(_LW_PENDING) >= _SIDE_LINEAR_GROUP_AT and _side_ok("linear"):
        _flush_linear_group(True)
    splits = int(_PURE.dfine_linear_wgrad_splits(M, N, K))
    np16, cp16 = _p16(N), _p16(K)
    return ws, (splits, N, K, 1, np16, cp16), (splits, N, 1, 1, np16, 1), splits * np16 * cp16


def _flush_conv_group(side=False):
    """The registered 1x1 weight gradients as ONE grouped launch (128 x 128 (n, c) tiles, dfine_conv_wgrad1_group)."""
    import numpy as np
    from .d_fine.arch.utils import upload
    pend = list(_CW_PENDING)
    _CW_PENDING.clear()
    if not pend:
        return
    table = np.empty((len(pend), 8), dtype=np.int64)
    blocks, flops, io = 1, 0.0, 0.0
    for i, (x, dy, ws, B, cin, cout, hw) in enumerate(pend):
        n = int(_lib.dfine_conv_wgrad1_group_row(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), B, cin, cout, hw, table[i].ctypes.data))
        if n < 0:
            raise RuntimeError("dfine_conv_wgrad1_group_row: bad arguments")
        blocks = max(blocks, n)
        flops += 2.0 * B * hw * cin * cout
        io += 2.0 * B * hw * (cin + cout) + 4.0 * cin * cout
    dev_table = upload(table, pend[0][0].device)
    if side:
        st = _side_fork(pend[0][0].device)         # (forked after the table's copy was enqueued)
        with _timed("conv1x1_wgrad", flops, io=io, stream=st.stream):
            _check(_lib.dfine_conv_wgrad1_group(_ptr(dev_table), len(pend), blocks, st.cuda_stream), "dfine_conv_wgrad1_group")
        _SIDE_LIVE.append((pend, dev_table))
        return
    with _timed("conv1x1_wgrad", flops, io=io):
        _check(_lib.dfine_conv_wgrad1_group(_ptr(dev_table), len(pend), blocks, _stream()), "dfine_conv_wgrad1_group")
    _LW_KEEP.append((pend, dev_table))


def _flush_linear_group(side=False):
    import numpy as np
    from .d_fine.arch.utils import upload
    pend = list(_LW_PENDING)
    _LW_PENDING.clear()
    table = np.empty((len(pend), 8), dtype=np.int64)
    blocks, flops, io = 1, 0.0, 0.0
    for i, (x2d, dy2d, ws, M, N, K) in enumerate(pend):
        n = int(_lib.dfine_linear_wgrad_group_row(_ptr(x2d), _ptr(dy2d), _ptr(ws), M, N, K, table[i].ctypes.data))
        if n < 0:
            raise RuntimeError("dfine_linear_wgrad_group_row: bad arguments")
        blocks = max(blocks, n)
        flops += 2.0 * M * N * K
        io += 2.0 * M * (N + K) + 4.0 * N * K
    dev_table = upload(table, pend[0][0].device)
    if side:
        st = _side_fork(pend[0][0].device)
        with _timed("linear_wgrad", flops, io=io, stream=st.stream):
            _check(_lib.dfine_linear_wgrad_group(_ptr(dev_table), len(pend), blocks, st.cuda_stream), "dfine_linear_wgrad_group")
        _SIDE_LIVE.append((pend, dev_table))
        return
    with _timed("linear_wgrad", flops, io=io):
        _check(_lib.dfine_linear_wgrad_group(_ptr(dev_table), len(pend), blocks, _stream()), "dfine_linear_wgrad_group")
    _LW_KEEP.append((pend, dev_table))          # inputs stay alive until the launch has run (stream order: dropped a few flushes later)


def linear_wgrad_flush(side=False):
    """Runs the registered weight-gradient problems (token-stream linears, 1x1 convolutions: partial sums into their `ws`
    buffers) in one launch per kind, then joins the side stream: afterwards every partial sum registered so far is ordered
    before whatever the current stream runs next.  side: the launches go to the side stream and nothing is joined (the
    caller continues there: an early reduction under the rest of backward)."""
    side = side and _side_ok()
    if _CW_PENDING:
        _flush_conv_group(side and _side_ok("group"))
    if _LW_PENDING:
        _flush_linear_group(side and _side_ok("linear"))
    while len(_LW_KEEP) > 4:
        _LW_KEEP.pop(0)
    if not side:
        side_join()


_LW_KEEP = []


def linear_wgrad_bf16(x2d, dy2d, with_bias=False, dw=None, db=None):
    """x2d [M, K], dy2d [M, N] bf16 contiguous -> dw [N, K] f32 = dy2d^T x2d (, db [N] f32 = column sums); `dw` / `db`
    may be given (contiguous row ranges of a larger gradient buffer)."""
    M, K = x2d.shape
    N = dy2d.shape[1]
    dev = x2d.device
    need = int(_PURE.dfine_linear_wgrad_ws_floats(M, N, K))
    key = (dev.index, _stream())
    ws = _LW_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), device=dev, dtype=torch.float32)
        _LW_WS[key] = ws
    if dw is None:
        dw = torch.empty(N, K, device=dev, dtype=torch.float32)
    if with_bias and db is None:
        db = torch.empty(N, device=dev, dtype=torch.float32)
    assert dw.is_contiguous() and dw.dtype == torch.float32 and (db is None or db.is_contiguous())
    with _timed("linear_wgrad", 2.0 * M * N * K, io=2.0 * M * (N + K) + 4.0 * N * K):
        _check(_lib.dfine_linear_wgrad_bf16(_ptr(x2d), _ptr(dy2d), _ptr(dw), _ptr(db), _ptr(ws), M, N, K, _stream()),
               "dfine_linear_wgrad_bf16")
    return (dw, db) if with_bias else dw


# ------------------------------------------------------------------------------------- HGNetv2 stem
def stem_supported(cin, cout, ks, stride):
    return bool(_PURE.dfine_stem_supported(cin, cout, ks, stride))


def stem_pack_weights(weight_f32, mode):
    """mode 0 forward, 1 stride-1 data gradient, 2 stride-2 data gradient (see dfine_hip.h)."""
    cout, cin, ks, _ = weight_f32.shape
    wp = torch.empty(weight_f32.numel(), device=weight_f32.device, dtype=torch.float32)
    _check(_lib.dfine_stem_pack_weights(_ptr(weight_f32), _ptr(wp), cout, cin, ks, mode, _stream()),
           "dfine_stem_pack_weights")
    return wp


def stem_conv(x, wp, cout, ks, stride, pad, out_hw):
    """x [B, Cin, H, W] bf16 contiguous -> y [B, cout, *out_hw] bf16 (zero fill outside the plane)."""
    B, cin, H, W = x.shape
    ho, wo = out_hw
    y = torch.empty(B, cout, ho, wo, device=x.device, dtype=torch.bfloat16)
    with _timed("stem_conv", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout)):
        _check(_lib.dfine_stem_conv_bf16(_ptr(x), _ptr(wp), _ptr(y), B, cin, cout, H, W, ho, wo, ks, stride, pad,
                                         _stream()), "dfine_stem_conv_bf16")
    return y


def stem_dgrad_s2(dy, wq, cin):
    B, cout, ho, wo = dy.shape
    dx = torch.empty(B, cin, 2 * ho, 2 * wo, device=dy.device, dtype=torch.bfloat16)
    with _timed("stem_conv", 2.0 * B * ho * wo * cin * cout * 9, io=2.0 * B * ho * wo * (cout + 4 * cin)):
        _check(_lib.dfine_stem_dgrad_s2_bf16(_ptr(dy), _ptr(wq), _ptr(dx), B, cin, cout, ho, wo, _stream()),
               "dfine_stem_dgrad_s2_bf16")
    return dx


_STEM_WS = {}


def _stem_pad32(dy):
    """The weight-gradient kernel walks the output rows in steps of 32 pixels: rows of another width run on a zero-padded
    copy of dy (zero gradients add nothing; the kernel bounds-checks its reads of x)."""
    wo = dy.shape[3]
    return dy if wo % 32 == 0 else torch.nn.functional.pad(dy, (0, 32 - wo % 32))


# The end of a captured backward: the main stream is done ~0.9 ms before the side stream (tools/stream_timeline.py: the last side
# graph - stage-1 3x3 weight gradient, the five stem weight gradients, the remaining grouped 1x1 / linear launches - only starts
# when the last main graph has ended).  So the registered groups are launched and the graph pair is closed when the backward pass
# reaches the stem: everything but the stem's own weight gradients runs under the stem's data gradients (28.39 -> 28.22 ms per
# step, same box, alternating; roofline fraction of the timed mode 0.127 -> 0.126).  Measured and dropped: every stem weight gradient
# in its own pair (28.13 ms, but those HBM-bound kernels then run beside the stem's HBM-bound main-stream kernels and both take
# longer: family kernel time +0.7 ms, fraction 0.123); a second cut in front of the third stem weight gradient (-0.065 ms, 0.1257).
def backward_tail_begins():
    """Called when the backward pass reaches the stem, its last stretch.  The stem's weight gradients are the end of the side
    stream's work and each can only start once the main stream has produced its dy, so whatever else is still queued for the
    side stream must not sit behind them: the registered grouped weight gradients (1x1 convolutions, token-stream linears) are
    launched NOW, under the stem's data gradients; and a captured backward (dl/engine.py: chain of (main, side) graph pairs, a
    side graph starts when its main graph has ended) closes its current pair at the next side launch (the first stem weight
    gradient): the side graph with the backlog starts there; the stem's own weight gradients keep one pair."""
    if not _side_ok():
        return
    if _CW_PENDING:
        _flush_conv_group(True)
    if _LW_PENDING:
        _flush_linear_group(True)
    if CAPTURE_DUAL is not None and CAPTURE_DUAL.cur is not None:
        n = getattr(CAPTURE_DUAL, "tail_calls", 0)
        CAPTURE_DUAL.tail_calls = n + 1
        if n == 0:
            CAPTURE_DUAL.cur[2] = max(CAPTURE_DUAL.cur[2], CAPTURE_DUAL.every)


def stem_wgrad(x, dy, ks, stride, pad, side=False):
    """side: launched on the side stream (see dwconv_backward)."""
    if side:
        backward_tail_begins()
    dy = _stem_pad32(dy)
    B, cin, H, W = x.shape
    _, cout, ho, wo = dy.shape
    need = int(_PURE.dfine_stem_wgrad_ws_floats(B, cin, cout, ks, ho, wo))
    side = side and _side_ok("stem")
    st = _side_fork(x.device, direct=True) if side else None
    key = (x.device.index, st.cuda_stream if side else _stream())
    ws = _STEM_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _STEM_WS[key] = torch.empty(need, device=x.device, dtype=torch.float32)
    dw = torch.empty(cout, cin, ks, ks, device=x.device, dtype=torch.float32)
    if side:
        with _timed("stem_wgrad", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout), stream=st.stream):
            _check(_lib.dfine_stem_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W, ho, wo, ks, stride,
                                              pad, st.cuda_stream), "dfine_stem_wgrad_bf16")
        _SIDE_LIVE.append((x, dy))
        return dw
    with _timed("stem_wgrad", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout)):
        _check(_lib.dfine_stem_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W, ho, wo, ks, stride,
                                          pad, _stream()), "dfine_stem_wgrad_bf16")
    return dw


def stem_conv2(xa, xb, wp, cout, ks, stride, pad, out_hw):
    """stem_conv of the channel concatenation [xa | xb] read in place."""
    B, ca, H, W = xa.shape
    cin = ca + xb.shape[1]
    ho, wo = out_hw
    y = torch.empty(B, cout, ho, wo, device=xa.device, dtype=torch.bfloat16)
    with _timed("stem_conv", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout)):
        _check(_lib.dfine_stem_conv2_bf16(_ptr(xa), _ptr(xb), ca, _ptr(wp), _ptr(y), B, cin, cout, H, W, ho, wo, ks, stride,
                                          pad, _stream()), "dfine_stem_conv2_bf16")
    return y


def stem_dgrad_s2_2(dy, wq, ca, cb):
    B, cout, ho, wo = dy.shape
    dxa = torch.empty(B, ca, 2 * ho, 2 * wo, device=dy.device, dtype=torch.bfloat16)
    dxb = torch.empty(B, cb, 2 * ho, 2 * wo, device=dy.device, dtype=torch.bfloat16)
    with _timed("stem_conv", 2.0 * B * ho * wo * (ca + cb) * cout * 9, io=2.0 * B * ho * wo * (cout + 4 * (ca + cb))):
        _check(_lib.dfine_stem_dgrad_s2_2_bf16(_ptr(dy), _ptr(wq), _ptr(dxa), _ptr(dxb), ca, B, ca + cb, cout, ho, wo,
                                               _stream()), "dfine_stem_dgrad_s2_2_bf16")
    return dxa, dxb


def stem_wgrad2(xa, xb, dy, ks, stride, pad, side=False):
    dy = _stem_pad32(dy)
    B, ca, H, W = xa.shape
    cin = ca + xb.shape[1]
    _, cout, ho, wo = dy.shape
    need = int(_PURE.dfine_stem_wgrad_ws_floats(B, cin, cout, ks, ho, wo))
    side = side and _side_ok("stem")
    if side:
        backward_tail_begins()
    st = _side_fork(xa.device, direct=True) if side else None
    key = (xa.device.index, st.cuda_stream if side else _stream())
    ws = _STEM_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _STEM_WS[key] = torch.empty(need, device=xa.device, dtype=torch.float32)
    dw = torch.empty(cout, cin, ks, ks, device=xa.device, dtype=torch.float32)
    if side:
        with _timed("stem_wgrad", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout), stream=st.stream):
            _check(_lib.dfine_stem_wgrad2_bf16(_ptr(xa), _ptr(xb), ca, _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W, ho, wo, ks,
                                               stride, pad, st.cuda_stream), "dfine_stem_wgrad2_bf16")
        _SIDE_LIVE.append((xa, xb, dy))
        return dw
    with _timed("stem_wgrad", 2.0 * B * ho * wo * cin * cout * ks * ks, io=2.0 * B * (H * W * cin + ho * wo * cout)):
        _check(_lib.dfine_stem_wgrad2_bf16(_ptr(xa), _ptr(xb), ca, _ptr(dy), _ptr(dw), _ptr(ws), B, cin, cout, H, W, ho, wo, ks,
                                           stride, pad, _stream()), "dfine_stem_wgrad2_bf16")
    return dw


def stem_pool_forward(x):
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    _check(_lib.dfine_stem_pool_fwd(_ptr(x), _ptr(y), B * C, H, W, _stream()), "dfine_stem_pool_fwd")
    return y


def stem_pool_backward(x, dy, acc=None):
    """acc: a gradient of x from its other consumer (bf16, contiguous): the pool's gradient is added onto it in place."""
    B, C, H, W = x.shape
    if acc is not None:
        _check(_lib.dfine_stem_pool_bwd_acc(_ptr(x), _ptr(dy), _ptr(acc), B * C, H, W, _stream()), "dfine_stem_pool_bwd_acc")
        return acc
    dx = torch.empty_like(x)
    _check(_lib.dfine_stem_pool_bwd(_ptr(x), _ptr(dy), _ptr(dx), B * C, H, W, _stream()), "dfine_stem_pool_bwd")
    return dx


# ------------------------------------------------------------------------------------- residual / gate + LayerNorm
def _dt(t):
    return 0 if t is None else _DTYPE[t.dtype]


def ln_fused_forward(mode, a, b, gate, weight, bias, eps, clampv, with_bf16=False):
    """a [.., D] (f32 / bf16), b same shape or None, gate [.., 2D] (mode 2) -> (y f32, mean, rstd[, y bf16 [rows, D]])."""
    D = a.shape[-1]
    rows = a.numel() // D
    y = torch.empty(a.shape, device=a.device, dtype=torch.float32)
    y16 = torch.empty(rows, D, device=a.device, dtype=torch.bfloat16) if with_bf16 else None
    mean = torch.empty(rows, device=a.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=a.device, dtype=torch.float32)
    _check(_lib.dfine_ln_fused_fwd(mode, _ptr(a), _dt(a), _ptr(b), _dt(b), _ptr(gate), _dt(gate), _ptr(weight), _ptr(bias),
                                   float(eps), float(clampv), _ptr(y), _ptr(y16), _ptr(mean), _ptr(rstd), rows, D, _stream()),
           "dfine_ln_fused_fwd")
    return (y, mean, rstd, y16) if with_bf16 else (y, mean, rstd)


def ln_fused_backward(mode, a, b, gate, weight, mean, rstd, dy, clampv, need_a, need_b, need_gate, need_affine, partials=False):
    """partials: the affine gradients are left as per-block partial sums - returns (da, db, dg, ws, blocks) with ws [blocks, 2, D]
    (row 0: weight, row 1: bias) for a deferred reduction instead of (da, db, dg, dweight, dbias)."""
    D = a.shape[-1]
    rows = a.numel() // D
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(b) if (need_b and b is not None) else None
    dg = torch.empty_like(gate) if (need_gate and gate is not None) else None
    dw = dbias = ws = None
    if need_affine:
        # two tensors, not two rows of one: AccumulateGrad adopts a whole tensor but copies a view
        if not partials:
            dw = torch.empty(D, device=a.device, dtype=torch.float32)
            dbias = torch.empty(D, device=a.device, dtype=torch.float32)
        ws = torch.empty(int(_PURE.dfine_ln_fused_bwd_ws_floats(rows, D)), device=a.device, dtype=torch.float32)
    _check(_lib.dfine_ln_fused_bwd(mode, _ptr(a), _dt(a), _ptr(b), _dt(b), _ptr(gate), _dt(gate), _ptr(weight), _ptr(mean),
                                   _ptr(rstd), _ptr(dy), float(clampv), _ptr(da), _ptr(db), _ptr(dg), _ptr(dw), _ptr(dbias),
                                   _ptr(ws), rows, D, _stream()), "dfine_ln_fused_bwd")
    if partials and need_affine:
        return da, db, dg, ws, ws.numel() // (2 * D)
    return da, db, dg, dw, dbias


# ------------------------------------------------------------------------------------- segmentation head (A10 / A15)
def groupnorm_forward(x, gamma, beta, groups, eps, relu):
    """x [B, C, H, W] f32 / bf16 contiguous -> (y, stat [B, G, 2])."""
    B, C = x.shape[:2]
    hw = x.numel() // max(B * C, 1)
    y = torch.empty_like(x)
    stat = torch.empty(B, groups, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(int(_PURE.dfine_groupnorm_ws_floats(B, C, groups)), device=x.device, dtype=torch.float32)
    _check(_lib.dfine_groupnorm_fwd(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(stat), _ptr(ws), _dtype_code(x), B, C, hw,
                                    groups, float(eps), int(bool(relu)), _stream()), "dfine_groupnorm_fwd")
    return y, stat


def groupnorm_backward(x, dy, gamma, beta, stat, groups, relu):
    """-> (dx, dgamma [C], dbeta [C])."""
    B, C = x.shape[:2]
    hw = x.numel() // max(B * C, 1)
    dx = torch.empty_like(x)
    part = torch.empty(B, C, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(B * groups * 2, device=x.device, dtype=torch.float32)
    _check(_lib.dfine_groupnorm_bwd(_ptr(x), _ptr(dy), _ptr(dx), _ptr(gamma), _ptr(beta), _ptr(stat), _ptr(part), _ptr(ws),
                                    _dtype_code(x), B, C, hw, groups, int(bool(relu)), _stream()), "dfine_groupnorm_bwd")
    s = part.sum(0)
    return dx, s[:, 1].contiguous(), s[:, 0].contiguous()


def bilinear_forward(x, out_hw, base=None):
    """x [..., Hi, Wi] contiguous -> [base +] resize to [..., Ho, Wo] (align_corners=False), a new tensor."""
    hi, wi = x.shape[-2:]
    planes = x.numel() // max(hi * wi, 1)
    y = torch.empty(*x.shape[:-2], out_hw[0], out_hw[1], device=x.device, dtype=x.dtype)
    _check(_lib.dfine_bilinear_fwd(_ptr(x), _ptr(base), _ptr(y), _dtype_code(x), planes, hi, wi, out_hw[0], out_hw[1],
                                   _stream()), "dfine_bilinear_fwd")
    return y


def bilinear_backward(dy, in_hw):
    ho, wo = dy.shape[-2:]
    planes = dy.numel() // max(ho * wo, 1)
    dx = torch.empty(*dy.shape[:-2], in_hw[0], in_hw[1], device=dy.device, dtype=dy.dtype)
    _check(_lib.dfine_bilinear_bwd(_ptr(dy), _ptr(dx), _dtype_code(dy), planes, in_hw[0], in_hw[1], ho, wo, _stream()),
           "dfine_bilinear_bwd")
    return dx


def mask_loss_sums(pm, plan_b, plan_q, plan_t, tgt, boxes):
    B, Q, H, W = pm.shape
    M = plan_b.numel()
    sums = torch.empty(M, 4, device=pm.device, dtype=torch.float32)
    _check(_lib.dfine_mask_loss_sums(_ptr(pm), _ptr(plan_b), _ptr(plan_q), _ptr(plan_t), _ptr(tgt), _ptr(boxes), _ptr(sums),
                                     _dtype_code(pm), M, Q, H, W, _stream()), "dfine_mask_loss_sums")
    return sums


def mask_loss_grad(pm, plan_b, plan_q, plan_t, tgt, boxes, coef):
    B, Q, H, W = pm.shape
    grad = torch.zeros_like(pm)
    _check(_lib.dfine_mask_loss_grad(_ptr(pm), _ptr(plan_b), _ptr(plan_q), _ptr(plan_t), _ptr(tgt), _ptr(boxes), _ptr(coef),
                                     _ptr(grad), _dtype_code(pm), plan_b.numel(), Q, H, W, _stream()), "dfine_mask_loss_grad")
    return grad


def mask_cost_sums(pm, gt, toff, q, tmax, alpha, gamma):
    """pm [B, Qall, H, W] logits, gt [sumT, H, W] f32, toff int32 [B + 1] (device) -> (out [B, q, tmax, 2], qsum [B, q, 2])."""
    B, qall, H, W = pm.shape
    out = torch.zeros(B, q, tmax, 2, device=pm.device, dtype=torch.float32)
    qsum = torch.empty(B, q, 2, device=pm.device, dtype=torch.float32)
    _check(_lib.dfine_mask_cost(_ptr(pm), _ptr(gt), _ptr(toff), _ptr(out), _ptr(qsum), _dtype_code(pm), B, qall, q, H * W, tmax,
                                float(alpha), float(gamma), _stream()), "dfine_mask_cost")
    return out, qsum


def conv1x1_batched_weights(x, w2, cout):
    """x [B, Cin, H, W] bf16, w2 [B, NP, KP] packed per image -> y [B, cout, H, W] bf16."""
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.bfloat16)
    with _timed("conv1x1", 2.0 * B * H * W * cin * cout, io=2.0 * B * H * W * (cin + cout) + 2.0 * cin * cout):
        _check(_lib.dfine_conv1x1_bw_bf16(_ptr(x), _ptr(w2), _ptr(y), B, cin, cout, H * W, _stream()), "dfine_conv1x1_bw_bf16")
    return y


# ------------------------------------------------------------------------------------- fp32 GEMMs (config #2)
def gemm_f32_nt(a, b, bias=None, alpha=1.0, act=0, splits=1, out=None):
    """a [..., M, K], b [..., N, K] fp32 with unit inner stride (leading batch dims equal, or b 2-D = shared) ->
    act(alpha * a @ b^T + bias) [..., M, N] fp32.  splits > 1: [batch * splits, M, N] partial products over K chunks (batch-major)."""
    if a.stride(-1) != 1:          # (size-1 inner dimensions carry arbitrary strides)
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    M, K = a.shape[-2], a.shape[-1]
    N = b.shape[-2]
    assert b.shape[-1] == K
    if a.dim() == 2:
        batch, sa, sb = 1, 0, 0
        a3, b3 = a, b
    else:
        a3 = a.reshape(-1, M, K) if a.dim() != 3 else a
        batch = a3.shape[0]
        if a3.stride(-1) != 1 or (M > 1 and a3.stride(1) < K):
            a3 = a3.contiguous()
        sa = a3.stride(0) if batch > 1 else 0
        if b.dim() == 2:
            b3, sb = b, 0
        else:
            b3 = b.reshape(-1, N, K) if b.dim() != 3 else b
            if b3.stride(-1) != 1:
                b3 = b3.contiguous()
            sb = b3.stride(0) if batch > 1 else 0
    lda = a3.stride(-2) if M > 1 else K
    ldb = b3.stride(-2) if N > 1 else K
    chunk = K
    if splits > 1:
        assert bias is None and act == 0
        chunk = ((K + splits - 1) // splits + 3) // 4 * 4
        splits = (K + chunk - 1) // chunk
    if out is None:
        shape = (batch * splits, M, N) if splits > 1 else (tuple(a.shape[:-2]) + (M, N))
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    with _timed("linear_f32", 2.0 * batch * M * N * K, io=4.0 * batch * (M * K + N * K + M * N), peak=F32_MFMA_PEAK_FLOPS):
        _check(_lib.dfine_gemm_f32_nt(a3.data_ptr(), b3.data_ptr(), _ptr(bias), out.data_ptr(), batch, M, N, K, lda, ldb, N, sa, sb,
                                      M * N, splits, chunk, float(alpha), int(act), _stream()), "dfine_gemm_f32_nt")
    return out


def gemm_f32(a, b, a_kmajor=False, b_kmajor=False, bias=None, alpha=1.0, act=0, splits=1):
    """C = act(alpha * op(a) @ op(b) + bias) in fp32 for 2-D or batched 3-D contiguous operands.  a is [.., M, K] (or [.., K, M] when
    a_kmajor), b is [.., N, K] (or [.., K, N] when b_kmajor; a 2-D b is shared by the batch).  splits > 1 (2-D operands): partial
    products [splits, M, N] over chunks of K."""
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    a = a if a.is_contiguous() else a.contiguous()
    b = b if b.is_contiguous() else b.contiguous()
    if a_kmajor:
        K, M = a.shape[-2], a.shape[-1]
    else:
        M, K = a.shape[-2], a.shape[-1]
    N = b.shape[-1] if b_kmajor else b.shape[-2]
    assert (b.shape[-2] if b_kmajor else b.shape[-1]) == K
    batch = 1 if a.dim() == 2 else a.numel() // (M * K)
    sa = M * K if batch > 1 else 0
    sb = N * K if (batch > 1 and b.dim() > 2) else 0
    chunk = K
    if splits > 1:
        assert batch == 1 and bias is None and act == 0
        chunk = ((K + splits - 1) // splits + 3) // 4 * 4
        splits = (K + chunk - 1) // chunk
    shape = (splits, M, N) if splits > 1 else (tuple(a.shape[:-2]) + (M, N))
    out = torch.empty(shape, device=a.device, dtype=torch.float32)
    with _timed("linear_f32", 2.0 * batch * M * N * K, io=4.0 * batch * (M * K + N * K + M * N), peak=F32_MFMA_PEAK_FLOPS):
        _check(_lib.dfine_gemm_f32(int(a_kmajor), int(b_kmajor), a.data_ptr(), b.data_ptr(), _ptr(bias), out.data_ptr(), batch, M, N, K,
                                   M if a_kmajor else K, N if b_kmajor else K, N, sa, sb, M * N, splits, chunk, float(alpha), int(act),
                                   _stream()), "dfine_gemm_f32")
    return out


def colsum_f32_ok(d):
    return d.dim() == 2 and d.dtype == torch.float32 and d.shape[1] % 4 == 0 and 4 <= d.shape[1] <= 1024 and d.shape[0] > 0


def colsum_f32(d, relu_y=None):
    """d [M, N] fp32 -> (part [splits, N] per-split column sums, dm): with relu_y [M, N] the sums are of dm = d * (relu_y > 0), which is
    returned as a new tensor; without it dm is d itself (made contiguous)."""
    d = d.contiguous()
    M, N = d.shape
    part = torch.empty(int(_lib.dfine_colsum_f32_splits(M)), N, device=d.device, dtype=torch.float32)
    dm = d
    if relu_y is not None:
        relu_y = relu_y.contiguous()
        dm = torch.empty_like(d)
    _check(_lib.dfine_colsum_f32(d.data_ptr(), _ptr(relu_y), _ptr(dm) if relu_y is not None else None, part.data_ptr(), M, N, _stream()),
           "dfine_colsum_f32")
    return part, dm


def planes_dense(x):
    """True for an NCHW tensor whose images are dense [C, H, W] blocks at any image stride - a contiguous map or a channel slice
    of one (x[:, a:b], a chunk of a concatenation or of its gradient): what the fp32 GEMM kernels read in place."""
    B, C, H, W = x.shape
    return x.stride(3) == 1 and x.stride(2) == W and x.stride(1) == H * W and (B == 1 or x.stride(0) >= C * H * W)


def conv1x1_f32(x, w2d):
    """x [B, Cin, H, W] fp32 (planes_dense: contiguous or a channel slice), w2d [Cout, Cin] fp32 contiguous -> [B, Cout, H, W]:
    y[b] = w2d @ x[b] (dfine_gemm_f32_nn)."""
    B, cin, H, W = x.shape
    cout = w2d.shape[0]
    hw = H * W
    if not planes_dense(x):
        x = x.contiguous()
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    with _timed("conv_f32", 2.0 * B * hw * cin * cout, io=4.0 * B * hw * (cin + cout), peak=F32_MFMA_PEAK_FLOPS):
        _check(_lib.dfine_gemm_f32_nn(w2d.data_ptr(), x.data_ptr(), None, y.data_ptr(), B, cout, hw, cin, cin, hw, hw, 0,
                                      x.stride(0) if B > 1 else cin * hw, cout * hw, 1.0, 0, _stream()), "dfine_gemm_f32_nn")
    return y


# ------------------------------------------------------------------------------------- evaluation masks (f2)
def mask_pack_bits(masks, thresh=0.5):
    """masks [N, H, W] uint8 (bit = value != 0) / float32 / bfloat16 (bit = value > thresh), contiguous -> [N, words] int64
    bit-packed masks (the layout is private to the library: only mask_iou_bits reads it)."""
    n, hw = masks.shape[0], int(masks.shape[-2]) * int(masks.shape[-1])
    dt = {torch.uint8: 0, torch.bool: 0, torch.float32: 1, torch.bfloat16: 2}[masks.dtype]
    words = int(_lib.dfine_mask_bits_words(hw))
    bits = torch.empty(n, words, device=masks.device, dtype=torch.int64)
    _check(_lib.dfine_mask_pack_bits(_ptr(masks.contiguous()), dt, float(thresh), n, hw, _ptr(bits), _stream()), "dfine_mask_pack_bits")
    return bits


def mask_iou_bits(pred_bits, gt_bits):
    """[Np, words], [Ng, words] packed masks of one image size -> [Np, Ng] float32 IoU."""
    iou = torch.zeros(pred_bits.shape[0], gt_bits.shape[0], device=pred_bits.device, dtype=torch.float32)
    _check(_lib.dfine_mask_iou_bits(_ptr(pred_bits), _ptr(gt_bits), pred_bits.shape[0], gt_bits.shape[0], pred_bits.shape[1],
                                    _ptr(iou), _stream()), "dfine_mask_iou_bits")
    return iou


# ------------------------------------------------------------------------------------- device data path (f3)
def mosaic_place(src, canvas, resized_hw, region, crop_xy):
    """src uint8 [Hs, Ws, 3], canvas uint8 [Hc, Wc, 3] (written in place): region (lx1, ly1, lx2, ly2) <- resize(src, resized_hw)
    cropped from crop_xy = (sx1, sy1)."""
    hs, ws = src.shape[:2]
    hc, wc = canvas.shape[:2]
    _check(_lib.dfine_mosaic_place_u8(_ptr(src), _ptr(canvas), hs, ws, int(resized_hw[0]), int(resized_hw[1]), hc, wc, int(region[0]),
                                      int(region[1]), int(region[2]), int(region[3]), int(crop_xy[0]), int(crop_xy[1]), _stream()),
           "dfine_mosaic_place_u8")


def warp_affine(src, m2x3, out_hw, border=114):
    """cv2.warpAffine(src uint8 [H, W, 3], M[:2], dsize=(out_w, out_h), borderValue=border) on the device."""
    hs, ws = src.shape[:2]
    dst = torch.empty(out_hw[0], out_hw[1], 3, device=src.device, dtype=torch.uint8)
    m = (ctypes.c_double * 6)(*[float(v) for v in m2x3.reshape(-1)])
    _check(_lib.dfine_warp_affine_u8(_ptr(src), _ptr(dst), hs, ws, out_hw[0], out_hw[1], m, int(border), _stream()), "dfine_warp_affine_u8")
    return dst


def affine_boxes(boxes, m2x3, scale, target_wh, area_thr):
    """boxes f32 [N, 4] xyxy (device) -> (transformed + clipped boxes [N, 4], keep u8 [N])."""
    boxes = boxes.float().contiguous()
    n = boxes.shape[0]
    out = torch.empty_like(boxes)
    keep = torch.empty(n, device=boxes.device, dtype=torch.uint8)
    m = (ctypes.c_float * 6)(*[float(v) for v in m2x3.reshape(-1)])
    _check(_lib.dfine_affine_boxes(_ptr(boxes), _ptr(out), _ptr(keep), n, m, float(scale), float(target_wh[0]), float(target_wh[1]),
                                   float(area_thr), _stream()), "dfine_affine_boxes")
    return out, keep


# ------------------------------------------------------------------------------------- fp32 convolutions (configs[1])
def conv_f32_pack_weights(w, dgrad):
    cout, cin, ks, _ = w.shape
    w2 = torch.empty(int(_lib.dfine_conv_f32_packed_elems(cout, cin, ks, int(dgrad))), device=w.device, dtype=torch.float32)
    _check(_lib.dfine_conv_f32_pack_weights(_ptr(w), _ptr(w2), cout, cin, ks, int(dgrad), _stream()), "dfine_conv_f32_pack_weights")
    return w2


def conv_f32_forward(x, w2, cout, ks, stride, pt, pl, out_hw):
    B, cin, hi, wi = x.shape
    y = torch.empty(B, cout, out_hw[0], out_hw[1], device=x.device, dtype=torch.float32)
    with _timed("conv_f32", 2.0 * B * out_hw[0] * out_hw[1] * cin * cout * ks * ks, io=4.0 * B * (hi * wi * cin + out_hw[0] * out_hw[1] * cout),
                peak=F32_MFMA_PEAK_FLOPS):
        _check(_lib.dfine_conv_f32_fwd(_ptr(x), _ptr(w2), _ptr(y), B, cin, cout, hi, wi, out_hw[0], out_hw[1], ks, stride, pt, pl, _stream()),
               "dfine_conv_f32_fwd")
    return y


def conv_f32_wgrad(x, dy, ks, stride, pt, pl, partials=False):
    """-> dw [Cout, Cin, ks, ks] f32, or (ws, meta) for the deferred reduction (meta = (splits, Cout, Cin, taps, NP16, CP16))."""
    B, cin, hi, wi = x.shape
    _, cout, ho, wo = dy.shape
    if ks == 1 and stride == 1 and cin % 16 == 0 and cout % 16 == 0 and (hi * wi) % 4 == 0:
        # 1x1: dW = sum over images and pixels of dY[b] X[b]^T - both operands pixel-contiguous, i.e. the NT GEMM itself with
        # (image, pixel chunk) as the split index (the generic kernel took 581 us per layer on D-FINE-s, 34 of a 99 ms step)
        hw = hi * wi
        tiles = ((cout + 63) // 64) * ((cin + 63) // 64)
        sp = max(1, min(hw // 256, -(-512 // (tiles * B))))
        part = gemm_f32_nt(dy.reshape(B, cout, hw), x.reshape(B, cin, hw), splits=sp)    # [B * splits, Cout, Cin] (slices read in place)
        if partials:
            return part.view(-1), (part.shape[0], cout, cin, 1, cout, cin)
        return part.sum(0).view(cout, cin, 1, 1)
    splits = int(_lib.dfine_conv_f32_wgrad_splits(B, cin, cout, ho, wo, ks))
    np16, cp16 = _p16(cout), _p16(cin)
    ws = torch.empty(splits * np16 * cp16 * ks * ks, device=x.device, dtype=torch.float32)
    with _timed("conv_f32", 2.0 * B * ho * wo * cin * cout * ks * ks, io=4.0 * B * (hi * wi * cin + ho * wo * cout), peak=F32_MFMA_PEAK_FLOPS):
        _check(_lib.dfine_conv_f32_wgrad(_ptr(x), _ptr(dy), _ptr(ws), B, cin, cout, hi, wi, ho, wo, ks, stride, pt, pl, _stream()),
               "dfine_conv_f32_wgrad")
    if partials:
        return ws, (splits, cout, cin, ks * ks, np16, cp16)
    return ws.view(splits, np16, cp16, ks * ks).sum(0)[:cout, :cin].reshape(cout, cin, ks, ks).contiguous()


def upsample2_zero(x, out_hw):
    planes = x.numel() // (x.shape[-1] * x.shape[-2])
    out = torch.empty(*x.shape[:-2], out_hw[0], out_hw[1], device=x.device, dtype=torch.float32)
    _check(_lib.dfine_upsample2_zero_f32(_ptr(x), _ptr(out), planes, x.shape[-2], x.shape[-1], out_hw[0], out_hw[1], _stream()),
           "dfine_upsample2_zero_f32")
    return out
