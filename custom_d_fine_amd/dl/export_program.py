"""(f4) Export artefact of the HIP-backed model: a `torch.export` ExportedProgram of the deployed eval forward + post-processor
(`ExportWrapper`, the module the reference hands to `torch.onnx.export`, src/dl/export.py:103-173) in which every C-ABI launch
is ONE opaque custom operator `dfine::call` - so the graph holds the ATen glue (views, casts, gathers, sigmoid ...) and the
launches of `libdfine_hip.so`, nothing else - saved with `torch.export.save` (`model.pt2`) and replayed by `load_program`.
The reference's ONNX -> TensorRT / OpenVINO tool-chains are NVIDIA / Intel products (and `onnx` is not in the image); the
ExportedProgram is the portable graph form this stack offers: a MIGraphX / ONNX-Runtime-ROCm lowering would start from it.

How the launches become graph nodes without twenty hand-written operator schemas: `dfine::call(Tensor?[] tensors, str spec)
-> Tensor[]` carries the name of a `custom_d_fine_amd.hip` wrapper and its non-tensor arguments as JSON; while exporting, the
wrappers of the eval path are replaced by shims that route through the operator.  Its fake (shape-inference) implementation runs
the REAL wrapper once on scratch tensors of the traced shapes / strides and reports the metadata of what came back - the GPU is
there at export time - so no shape function can drift from the kernels.
"""
import contextlib
import json
from pathlib import Path
from typing import List, Optional

import torch

from .. import hip, kernels

_SPEC_OUT = {}            # spec string -> structure of the wrapper's return value ("t" | "n" | ["t", "n", ...])
# hip wrappers of the eval path (forward only).  (name, index of an `out=`-style argument the wrapper fills in place or None)
_EVAL_WRAPPERS = ("conv_pack_weights", "conv_forward_bf16", "bn_act_forward", "bn2_act_forward", "dwconv_forward", "stem_pack_weights",
                  "stem_conv", "stem_conv2", "stem_pool_forward", "maps_to_tokens", "upsample2_nearest", "act_forward", "attn_forward",
                  "ln_fused_forward", "msda_fused_forward", "fdr_forward", "topk_anchors", "postprocess", "groupnorm_forward",
                  "bilinear_forward", "conv1x1_batched_weights", "gemm_f32", "gemm_f32_nt", "conv1x1_f32", "conv_f32_forward",
                  "conv_f32_pack_weights", "conv_forward_affine", "conv1x1_seg_forward_affine", "bn_fold", "dwconv_forward_affine")


def _encode(args, kwargs):
    """(args, kwargs) of a hip wrapper -> (tensor list, JSON spec of everything else with tensor references)."""
    tensors = []

    def enc(v):
        if torch.is_tensor(v):
            tensors.append(v)
            return {"t": len(tensors) - 1}
        if isinstance(v, (list, tuple)):
            return {"l": [enc(x) for x in v], "tuple": isinstance(v, tuple)}
        if isinstance(v, torch.dtype):
            return {"dtype": str(v).split(".")[-1]}
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        raise TypeError(f"export: cannot serialise argument of type {type(v)}")
    return tensors, {"a": [enc(v) for v in args], "k": {k: enc(v) for k, v in kwargs.items()}}


def _decode(node, tensors):
    if isinstance(node, dict):
        if "t" in node:
            return tensors[node["t"]]
        if "dtype" in node:
            return getattr(torch, node["dtype"])
        if "l" in node:
            seq = [_decode(x, tensors) for x in node["l"]]
            return tuple(seq) if node["tuple"] else seq
    return node


def _flatten_out(res):
    if res is None:
        return [], "n"
    if torch.is_tensor(res):
        return [res], "t"
    flat, struct = [], []
    for r in res:
        if r is None:
            struct.append("n")
        else:
            assert torch.is_tensor(r), "export: nested wrapper results are not supported"
            flat.append(r)
            struct.append("t")
    return flat, struct


def _unflatten_out(flat, struct):
    if struct == "n":
        return None
    if struct == "t":
        return flat[0]
    it = iter(flat)
    return tuple(next(it) if s == "t" else None for s in struct)


def _run(tensors, spec):
    s = json.loads(spec)
    fn = _ORIGINAL[s["fn"]]
    res = fn(*[_decode(a, tensors) for a in s["a"]], **{k: _decode(v, tensors) for k, v in s["k"].items()})
    flat, struct = _flatten_out(res)
    _SPEC_OUT[spec] = struct
    # an operator's outputs must not alias its inputs
    ins = {t.data_ptr() for t in tensors if t is not None and t.numel()}
    return [o.clone() if (o.numel() and o.data_ptr() in ins) else o for o in flat]


@torch.library.custom_op("dfine::call", mutates_args=())
def dfine_call(tensors: List[Optional[torch.Tensor]], spec: str) -> List[torch.Tensor]:
    return _run(tensors, spec)


@dfine_call.register_fake
def _(tensors, spec):
    from torch._subclasses.fake_tensor import unset_fake_temporarily
    metas = [None if t is None else (tuple(t.shape), tuple(t.stride()), t.dtype, t.device) for t in tensors]
    with unset_fake_temporarily():
        real = [None if m is None else torch.empty_strided(m[0], m[1], dtype=m[2], device=m[3]).zero_() for m in metas]
        outs = _run(real, spec)
        out_meta = [(tuple(o.shape), tuple(o.stride()), o.dtype, o.device) for o in outs]
        del real, outs
    return [torch.empty_strided(sh, st, dtype=dt, device=dv) for sh, st, dt, dv in out_meta]


_ORIGINAL = {name: getattr(hip, name) for name in _EVAL_WRAPPERS if hasattr(hip, name)}
_ORIGINAL["linear_act"] = hip.linear_act
_ORIGINAL["conv1x1_seg_new"] = None          # filled below (functional form of conv1x1_seg_forward)


_SEG_FORWARD = hip.conv1x1_seg_forward      # (the module attribute is a shim while exporting)


def _conv1x1_seg_new(x_parts, w2, shape, like):
    y = torch.empty(shape, device=like.device, dtype=torch.bfloat16)
    _SEG_FORWARD(tuple(x_parts), w2, (y,))
    return y


_ORIGINAL["conv1x1_seg_new"] = _conv1x1_seg_new


def _shim(name):
    def call(*args, **kwargs):
        tensors, spec = _encode(args, kwargs)
        spec["fn"] = name
        key = json.dumps(spec, sort_keys=True)
        flat = torch.ops.dfine.call(tensors, key)
        return _unflatten_out(list(flat), _SPEC_OUT[key])
    return call


def _is_channel_part_by_strides(t):
    if t.dim() != 4 or t.dtype != torch.bfloat16:
        return False
    B, C, H, W = t.shape
    hw = H * W
    return (t.stride(3) == 1 or W == 1) and (t.stride(2) == W or H == 1) and (t.stride(1) == hw or C == 1) and (
        B == 1 or (t.stride(0) % hw == 0 and t.stride(0) >= C * hw)) and (t.storage_offset() * 2) % 16 == 0


@contextlib.contextmanager
def _export_mode(model):
    """Routes the eval path's launches through `dfine::call`, the weight caches through plain operators (an exported graph
    re-derives packed / bf16 weights from its parameters at every run) and the decoder's host constants through values fetched
    before tracing."""
    saved_hip = {n: getattr(hip, n) for n in _ORIGINAL if hasattr(hip, n)}
    saved = (kernels._packed_weights, kernels.bf16_param, kernels._packed_stem, kernels._packed_f32, hip.is_channel_part,
             hip.conv1x1_seg_forward)
    call_pack, call_stem, call_f32 = _shim("conv_pack_weights"), _shim("stem_pack_weights"), _shim("conv_f32_pack_weights")
    seg_new = _shim("conv1x1_seg_new")
    lin = _shim("linear_act")

    def linear_act(x2d, w, bias=None, act=0, out_f32=False, out=None):
        res = lin(x2d, w, bias, act, out_f32)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def seg_forward(x_parts, w2, y_parts):
        if len(y_parts) != 1:
            raise RuntimeError("export: a segmented 1x1 convolution with several outputs only occurs in backward")
        y_parts[0].copy_(seg_new(list(x_parts), w2, list(y_parts[0].shape), x_parts[0]))

    dec = getattr(model, "decoder", None)
    fdr_saved = None
    if dec is not None and hasattr(dec, "decoder") and hasattr(dec.decoder, "_fdr_constants"):
        inner = dec.decoder
        fdr_saved = inner._fdr_constants
        project = inner.project if hasattr(inner, "project") else None
        consts = {}
        if project is not None:
            consts["v"] = fdr_saved(project, dec.reg_scale)
        inner._fdr_constants = lambda project_, reg_scale_: consts["v"] if "v" in consts else fdr_saved(project_, reg_scale_)
    try:
        for n in saved_hip:
            if n not in ("linear_act",):
                setattr(hip, n, _shim(n))
        hip.linear_act = linear_act
        hip.conv1x1_seg_forward = seg_forward
        hip.is_channel_part = _is_channel_part_by_strides
        kernels._packed_weights = lambda w, dgrad: call_pack(w.detach().float().contiguous(), dgrad)
        kernels._packed_stem = lambda w, mode: call_stem(w.detach().float().contiguous(), mode)
        kernels._packed_f32 = lambda w, dgrad: call_f32(w.detach().float().contiguous(), dgrad)
        kernels.bf16_param = lambda p: p.detach().to(torch.bfloat16)
        yield
    finally:
        for n, f in saved_hip.items():
            setattr(hip, n, f)
        (kernels._packed_weights, kernels.bf16_param, kernels._packed_stem, kernels._packed_f32, hip.is_channel_part,
         hip.conv1x1_seg_forward) = saved
        if fdr_saved is not None:
            dec.decoder._fdr_constants = fdr_saved


class _Half(torch.nn.Module):
    """The wrapper under bf16 autocast (what `Torch_model(half=True)` runs), as a module so that the context is part of the trace."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x):
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            return self.inner(x)


def export_program(model, num_classes, img_size, path, batch=1, half=True, deploy=True):
    """Writes `path` (.pt2): ExportedProgram of post-processor(model(x)) for x [batch, 3, H, W] on the model's device.
    -> (path, ExportedProgram).  `model` is put into eval mode (and re-parameterised with `deploy()` unless deploy=False)."""
    from .export import DFINEPostProcessor, ExportWrapper
    if not deploy:
        raise NotImplementedError("export_program exports the deployed (re-parameterised) model, like Torch_model serves it")
    model = model.eval()
    model.deploy()
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("export_program traces the HIP-backed forward: the model must live on the GPU")
    wrapper = ExportWrapper(model, DFINEPostProcessor(num_classes), tuple(img_size)).eval()
    mod = _Half(wrapper) if half else wrapper
    x = torch.rand(batch, 3, img_size[0], img_size[1], device=dev)
    with torch.no_grad():
        mod(x)                                  # fills the shape-keyed caches (anchors, routes) with real tensors first
        with _export_mode(model):
            ep = torch.export.export(mod, (x,), strict=False)
    path = Path(path)
    torch.export.save(ep, str(path))
    return path, ep


def load_program(path):
    """-> callable(x) replaying the exported graph through libdfine_hip.so (the `dfine::call` operator is registered by importing
    this module)."""
    ep = torch.export.load(str(path))
    return ep.module()


def graph_targets(ep):
    """Targets of every call_function node of the program, nested graphs (the autocast region is one) included."""
    out = []
    for mod in ep.graph_module.modules():
        g = getattr(mod, "graph", None)
        if g is not None:
            out += [str(n.target) for n in g.nodes if n.op == "call_function"]
    return out


def count_launch_nodes(ep):
    return sum(1 for t in graph_targets(ep) if "dfine.call" in t)
