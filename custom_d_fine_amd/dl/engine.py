"""One optimisation step of the hot path and its satellites (EMA, data-parallel wrap).

Restates the step semantics of the reference's `Trainer.train` / `optimizer_step`
(`src/dl/train.py:512-535,550-586`): autocast forward, fp32 loss, `sum(loss_dict)/accum`,
backward, [unscale ->] clip_grad_norm_(max_norm) -> AdamW -> scheduler -> zero_grad -> EMA.
bf16 autocast needs no GradScaler (the reference's fp16 path does).
"""
import math
import os
from copy import deepcopy

import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from ..d_fine.dist_utils import get_world_size, is_dist_available_and_initialized


class ModelEMA:
    """Exponential moving average over every floating-point state-dict entry (parameters AND
    BatchNorm statistics), momentum m * (1 - exp(-iters / 2000))  (ref train.py:52-73).
    The reference issues two tiny kernels per tensor (2106 launches for D-FINE-m); here the
    update is two multi-tensor launches."""

    def __init__(self, student, ema_momentum):
        student = student.module if isinstance(student, DDP) else student
        self.model = deepcopy(student).eval()
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.ema_scheduler = lambda x: ema_momentum * (1 - math.exp(-x / 2000))
        self._pairs = None

    def _tensor_pairs(self, student):
        if self._pairs is None or self._pairs[0] is not student:
            ema_sd, stu_sd = self.model.state_dict(), student.state_dict()
            names = [k for k, v in ema_sd.items() if v.dtype.is_floating_point]
            self._pairs = (student, [ema_sd[k] for k in names], [stu_sd[k] for k in names])
        return self._pairs[1], self._pairs[2]

    @torch.no_grad()
    def update(self, iters, student):
        student = student.module if isinstance(student, DDP) else student
        momentum = self.ema_scheduler(iters)
        ema, stu = self._tensor_pairs(student)
        torch._foreach_mul_(ema, momentum)
        torch._foreach_add_(ema, stu, alpha=1.0 - momentum)


def wrap_data_parallel(model, device):
    """One process per GPU; gradients are averaged by bucketed all-reduce (RCCL over xGMI on
    ROCm) overlapped with backward.  D-FINE-m carries 78 MB of fp32 gradients: with xGMI's
    point-to-point links a ring step is per-link bound, so few large buckets beat many small
    ones - 2 x ~40 MB here instead of DDP's default 25 MB.  BatchNorm buffers are per-rank
    statistics (the reference's DDP default re-broadcasts rank 0's every forward; dropped)."""
    if not is_dist_available_and_initialized() or get_world_size() == 1:
        return model
    if device.type == "cuda":
        return DDP(model, device_ids=[device.index], output_device=device.index,
                   find_unused_parameters=False, broadcast_buffers=False,
                   gradient_as_bucket_view=True, bucket_cap_mb=40)
    return DDP(model, find_unused_parameters=False, broadcast_buffers=False)


class _BackboneEncoder(nn.Module):
    def __init__(self, backbone, encoder):
        super().__init__()
        self.backbone, self.encoder = backbone, encoder

    def forward(self, x):
        return tuple(self.encoder(self.backbone(x)))


class _Backbone(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, x):
        return tuple(self.backbone(x))


class _Encoder(nn.Module):
    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder

    def forward(self, *feats):
        return tuple(self.encoder(list(feats)))


class _DualCapture:
    """Backward capture as a chain of graph PAIRS: pair k = (main_k, side_k); main_k holds what the backward pass launches
    on its own stream (BatchNorm backward -> data gradient -> ...), side_k what it launches on the side stream during the
    same stretch (weight gradients).  Replay: main_0, [event] side_0 || main_1, [event] side_1 || main_2 ... join.  Every
    launch of side_k reads results of main_j, j <= k, only (it was issued after them), so the one event per pair orders it.
    Why not fork the side stream INTO one capture: the runtime then executes the branches without any overlap (measured,
    tools/graph_probe8.py: 17.2 ms with or without the side stream inside the graph, against 15.9 ms for the eager launch
    sequence; DEBUG_HIP_FORCE_GRAPH_QUEUES 1 / 2 / 8 change nothing)."""

    def __init__(self, cap_stream, side, pool, every):
        self.cap_stream, self.side, self.pool, self.every = cap_stream, side, pool, max(int(every), 1)
        self.side_pool = torch.cuda.graph_pool_handle()      # two captures that are open at the same time cannot share a pool
        self.pairs, self.n, self.cur = [], 0, None
        self.deferred = []          # per pair: its side graph holds launches that write partial sums for the deferred split reduction

    def begin(self):
        gm, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.cap_stream):
            gm.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        with torch.cuda.stream(self.side.stream):
            gs.capture_begin(pool=self.side_pool, capture_error_mode="relaxed")
        self.cur = [gm, gs, 0, False]

    def end(self):
        import warnings
        gm, gs, n, deferred = self.cur
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                  # "The CUDA Graph is empty": a pair without side-stream launches
            with torch.cuda.stream(self.side.stream):
                gs.capture_end()
            with torch.cuda.stream(self.cap_stream):
                gm.capture_end()
        self.pairs.append((gm, gs if n else None))
        self.deferred.append(bool(n and deferred))
        self.cur = None

    def side_launch(self, direct=False):
        if self.cur[2] >= self.every:           # this launch opens the next pair
            self.end()
            self.begin()
        self.cur[2] += 1
        if not direct:
            self.cur[3] = True


class GraphedSegment:
    """A module with static shapes (backbone + encoder) whose training forward and backward are each captured ONCE into a HIP
    graph and replayed afterwards: two graph launches (~0.1 ms of host time each, measured: tools/graph_probe7.py) instead
    of ~1 000 kernel launches issued from ~700 Python autograd nodes (~16 ms of host time per step for D-FINE-m - the step
    was host-paced).  The device-side work is the eager path's, launch for launch:

    * the parameters are views of the fused optimizer's flat buffer (fixed addresses); their packed / bf16 copies are the
      registered shadows of `kernels`, refreshed with one launch per kind in front of every forward replay;
    * the backward graph ends with the segment's gradients IN the flat gradient buffer: the deferred weight-gradient
      partial sums are reduced there by the same grouped launches as in the eager step (dfine_conv_wgrad1_group,
      dfine_linear_wgrad_group, dfine_multi_wgrad_reduce), gradient tensors (BatchNorm affine, depthwise, stem ...) are added
      by one dfine_multi_add_f32 launch - no autograd leaf of the model takes part in a replay;
    * the weight-gradient launches keep their side stream: the backward pass is recorded as a chain of graph pairs
      (_DualCapture: what goes to the main stream / to the side stream during a stretch of 5 weight-gradient launches),
      replayed pair by pair on the two streams (`DFINE_GRAPH_SIDE=fork`: the side stream forked into ONE capture - the
      runtime then runs the branches without overlap; `DFINE_GRAPH_SIDE=0`: one stream);
    * BatchNorm running statistics are updated by the recorded kernels; the `num_batches_tracked` counters by the same
      deferred bookkeeping as in the eager step.

    The capture runs the module on ALIASES of the parameters (detached views made on the capture stream): autograd ties a
    leaf's gradient accumulator to the stream it was created on, and an accumulator left over from an eager step (default
    stream, kept alive by any graph the caller still holds) makes the engine synchronise the capturing stream with the
    legacy stream - a crash on this stack (tools/graph_probe6.py).  `torch.cuda.make_graphed_callables` segfaults here even
    for a two-layer MLP (tools/graph_probe2.py)."""

    GROUP_AT = 16     # registered 1x1 / linear weight gradients per grouped launch inside a captured backward
    CHUNK = 8         # side-stream launches per (main, side) graph pair (2 / 3 / 8: +0.2 / +0.2 / 0 ms per step in round 5)
    # (8, 5) until the end of round 6; (16, 8): 26.91 - 26.97 -> 26.80 - 26.83 ms per step, four same-box alternations of the two builds
    # (tools/ab_trees.sh; in-process with AB_RECAPTURE=1 each of the two alone -0.04 .. -0.27)

    def __init__(self, module, sample_inputs, amp_dtype=None, fused=None, warmup=1, input_grads=False, clone_inputs=True,
                 defer_backward=False, cast_inputs=False):
        """cast_inputs: floating-point inputs are KEPT in amp_dtype - the per-replay copy into the static input is the cast the
        module's first layer would do (the stem convolution reads bf16: one pass over the images instead of a same-type copy plus a
        cast inside the graph; only for inputs that need no gradient and that the module reads through autocast ops only).
        input_grads: the segment is not the first one of the model - its backward also produces the gradients of its
        (floating-point) inputs, static tensors too (`static_gin`).  clone_inputs=False: the inputs ARE static tensors already (the
        outputs of the segment in front: no copy per replay).  defer_backward: only the forward is captured here; the caller
        finishes with `capture_backward(static_gout)` once it can say where the output gradients will be written (the
        `static_gin` of the segment behind), so that no gradient map is copied between two segments."""
        from .. import hip, kernels
        from ..d_fine.arch import utils as arch_utils
        if fused is None:
            raise RuntimeError("GraphedSegment delivers its gradients into FusedAdamWEMA's flat gradient buffer")
        self.module, self.fused, self.amp_dtype = module, fused, amp_dtype
        self.hip, self.kernels = hip, kernels
        dev = sample_inputs[0].device
        self.device = dev
        self.input_grads = input_grads
        self.done_before = ()        # top-level modules whose backward is over when this segment's starts (set by the caller)
        self.static_in = [t.detach().clone() if clone_inputs else t.detach() for t in sample_inputs]
        if cast_inputs and amp_dtype is not None and clone_inputs and not input_grads:
            self.static_in = [t.to(amp_dtype) if t.dtype == torch.float32 else t for t in self.static_in]
        self.static_gin = None
        self._keep, self._pinned = [], hip.CaptureArena()
        self.side = os.environ.get("DFINE_GRAPH_SIDE", "dual")           # "dual" | "fork" | "0"
        if not hip.WGRAD_STREAM:
            self.side = "0"

        cap_stream = torch.cuda.Stream(device=dev)
        cap_stream.wait_stream(torch.cuda.current_stream(dev))
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        with torch.cuda.stream(cap_stream):
            aliases = []
            for _, p in named:
                a = p.detach().requires_grad_(True)
                slot = getattr(p, "_dfine_slot", None)
                if slot is not None:
                    a._dfine_slot = slot
                aliases.append(a)
            if input_grads:                          # leaves made on the capture stream, like the parameter aliases
                self.static_in = [t.requires_grad_(True) if t.dtype.is_floating_point else t for t in self.static_in]
        self._gin_of = [i for i, t in enumerate(self.static_in) if input_grads and t.dtype.is_floating_point]
        wrt_inputs = [self.static_in[i] for i in self._gin_of]
        alias_map = {n: a for (n, _), a in zip(named, aliases)}
        self._aliases = aliases
        self.param_index = [getattr(p, "_dfine_slot", (None, None))[1] for _, p in named]
        if any(i is None for i in self.param_index) or any(getattr(p, "_dfine_slot")[0] is not fused for _, p in named):
            raise RuntimeError("every trainable parameter of a graphed segment must live in the fused optimizer's flat buffers")

        def run():
            with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None, cache_enabled=False):
                return tuple(torch.func.functional_call(module, alias_map, tuple(self.static_in)))

        def deliver(grads):
            """Segment gradients -> flat gradient buffer (recorded at the end of the backward capture)."""
            hip.linear_wgrad_flush()                    # grouped weight-gradient launches of what is registered; joins the side stream
            real = [(g, fused.grad_offset(i)) for g, i in zip(grads, self.param_index) if g is not None]
            for g, _ in real:
                if g.dtype != torch.float32:
                    raise RuntimeError("graphed segment: parameter gradients are expected in fp32")
            if real:
                self._keep.append(hip.multi_copy_f32([g.contiguous() for g, _ in real], [o for _, o in real], fused.flat_grad, add=True))
            fused._flush_deferred()                     # the deferred partial sums of the segment, reduced into flat_grad

        if fused._deferred or hip._CW_PENDING or hip._LW_PENDING:
            raise RuntimeError("GraphedSegment must be built between steps (weight gradients of a running backward are pending)")
        was_accumulating = fused.accumulating
        fused.accumulating = True                       # no bucket bookkeeping from the warm-up / capture runs
        snap_grad = fused.flat_grad.clone()
        snap_buf = None if fused.flat_buf is None else fused.flat_buf.clone()
        # the warm-up is a real training-mode forward: BatchNorm running statistics / counters that do not live in the
        # optimizer's flat buffer (no EMA -> no flat_buf; integer counters never do) are rolled back module by module
        snap_mod = [(b, b.clone()) for b in module.buffers() if fused.flat_buf is None or not b.dtype.is_floating_point]
        snap_bn = dict(kernels._BN_PENDING)
        flags = (kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS, hip.CAPTURE_SIDE, arch_utils.CAPTURE_KEEP)
        self.bwd_pairs = None
        group_at = hip._SIDE_GROUP_AT
        # grouped weight-gradient launches every few registrations: a replay has no host cost per launch (the eager step
        # groups 32 to save ~25 us of host time each), and small groups keep the side stream's work evenly spread
        # (4 / 16 / 24 / 32 problems instead of 8: +0.05 / -0.05 / -0.08 / 0 ms per step, roofline fraction 0.122-0.125 against 0.127)
        hip._SIDE_GROUP_AT = self.GROUP_AT
        try:
            # ---- eager warm-up on the capture stream: fills the shadow registries for the aliases, sizes the workspaces
            with torch.cuda.stream(cap_stream):
                for _ in range(max(warmup, 1)):
                    outs = run()
                    grads = torch.autograd.grad(outs, aliases + wrt_inputs, [torch.ones_like(o) for o in outs], allow_unused=True)
                    deliver(grads[:len(aliases)])
                    del outs, grads
            cap_stream.synchronize()
            fused.flat_grad.copy_(snap_grad)
            if snap_buf is not None:
                fused.flat_buf.copy_(snap_buf)
            for b, c in snap_mod:
                b.copy_(c)
            kernels._BN_PENDING.clear()
            kernels._BN_PENDING.update(snap_bn)
            fused._uses.clear()
            self._keep.clear()
            torch.cuda.synchronize(dev)

            kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS = True, True
            hip.CAPTURE_SIDE = self.side == "fork"
            arch_utils.CAPTURE_KEEP = self._pinned
            kernels.refresh_weight_shadows(dev)
            torch.cuda.synchronize(dev)
            self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd_graph, stream=cap_stream, capture_error_mode="relaxed"):
                self.static_out = run()
                hip.side_join()
            self._bn_counters = [(k, v[0]) for k, v in kernels._BN_PENDING.items() if snap_bn.get(k, (None, 0))[1] != v[1]]
            kernels._BN_PENDING.clear()
            kernels._BN_PENDING.update(snap_bn)
        finally:
            kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS, hip.CAPTURE_SIDE, arch_utils.CAPTURE_KEEP = flags
            hip._SIDE_GROUP_AT = group_at
            fused.accumulating = was_accumulating

        def capture_backward(static_gout=None):
            """Second half of the construction: the backward pass as a chain of (main, side) graph pairs + the delivery graph.
            static_gout: the tensors the output gradients arrive in (default: own zero-filled buffers)."""
            flags_b = (kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS, hip.CAPTURE_SIDE, arch_utils.CAPTURE_KEEP)
            group_at_b, was_acc = hip._SIDE_GROUP_AT, fused.accumulating
            hip._SIDE_GROUP_AT = self.GROUP_AT
            fused.accumulating = True
            if fused._deferred or hip._CW_PENDING or hip._LW_PENDING:
                raise RuntimeError("GraphedSegment.capture_backward: weight gradients of another backward are pending")
            try:
                kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS = True, True
                hip.CAPTURE_SIDE = self.side == "fork"
                arch_utils.CAPTURE_KEEP = self._pinned
                self.static_gout = list(static_gout) if static_gout is not None else [torch.zeros_like(o) for o in self.static_out]
                for o, g in zip(self.static_out, self.static_gout):
                    if g.shape != o.shape or g.dtype != o.dtype or not g.is_contiguous():
                        raise RuntimeError("static_gout must match the segment's outputs (shape, dtype, contiguous)")
                torch.cuda.synchronize(dev)
                if self.side == "dual":
                    # the backward pass proper as a chain of (main, side) graph pairs, then the delivery of the gradients as one graph
                    dual = _DualCapture(cap_stream, hip.side_stream(dev), self.fwd_graph.pool(), self.CHUNK)
                    torch.cuda.synchronize(dev)
                    hip.CAPTURE_DUAL = dual
                    try:
                        with torch.cuda.stream(cap_stream):
                            dual.begin()
                            grads = torch.autograd.grad(self.static_out, aliases + wrt_inputs, self.static_gout, allow_unused=True)
                            hip.linear_wgrad_flush(side=True)      # what is still registered: grouped launches on the side stream
                            dual.end()
                    finally:
                        hip.CAPTURE_DUAL = None
                    self.bwd_pairs = dual.pairs
                    self._keep.append(list(hip._SIDE_LIVE))       # inputs of the side graphs: referenced until every capture is done
                    hip._SIDE_LIVE.clear()
                    with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool(), stream=cap_stream, capture_error_mode="relaxed"):
                        deliver(grads[:len(aliases)])
                else:
                    with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool(), stream=cap_stream, capture_error_mode="relaxed"):
                        grads = torch.autograd.grad(self.static_out, aliases + wrt_inputs, self.static_gout, allow_unused=True)
                        deliver(grads[:len(aliases)])
                self._static_grads = grads
                if self._gin_of:
                    gin = [None] * len(self.static_in)
                    for i, g in zip(self._gin_of, grads[len(aliases):]):
                        if g is None:
                            raise RuntimeError("graphed segment: an input that requires a gradient got none")
                        gin[i] = g.contiguous()
                    self.static_gin = gin
                fused._uses.clear()
                # scratch buffers the recorded launches point into live in module-level tables keyed by stream / shape: held
                # here so that a later, larger request cannot free them under the graph
                self._keep.append([list(d.values()) for d in (hip._BN_WS, hip._LW_WS, hip._STEM_WS)])
                self._keep.append(list(fused._live))
            finally:
                kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_SHADOWS, hip.CAPTURE_SIDE, arch_utils.CAPTURE_KEEP = flags_b
                hip.CAPTURE_DUAL = None
                hip._SIDE_GROUP_AT, fused.accumulating = group_at_b, was_acc
            self.capture_backward = None

        self.capture_backward = capture_backward
        if not defer_backward:
            capture_backward()
        self._token = torch.zeros((), device=dev, requires_grad=True)     # makes autograd call _Replay.backward
        seg = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, token, *args):
                for dst, src in zip(seg.static_in, args):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                kernels.refresh_weight_shadows(seg.device)
                seg.fwd_graph.replay()
                if kernels._BN_DEFER:
                    pend = kernels._BN_PENDING
                    for key, buf in seg._bn_counters:
                        ent = pend.get(key)
                        pend[key] = (buf, 1 if ent is None else ent[1] + 1)
                else:
                    torch._foreach_add_([b for _, b in seg._bn_counters], 1)
                outs = tuple(o.detach() for o in seg.static_out)
                for o, buf in zip(outs, seg.static_gout):
                    o._dfine_grad_buf = buf          # consumers that can (kernels.flatten_levels) write the gradient there
                return outs

            @staticmethod
            def backward(ctx, *gouts):
                for dst, src in zip(seg.static_gout, gouts):
                    if src is None:
                        dst.zero_()
                    elif dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                f = seg.fused
                if seg.done_before:
                    f.module_backward_done(seg.done_before)     # e.g. the decoder's buckets, whatever parameter got no gradient
                if not f.accumulating and hip.side_stream_ok():
                    # what the decoder's backward has registered so far - grouped weight-gradient launches still pending, partial
                    # sums to reduce - goes to the side stream now, under the segment's backward, instead of running serially
                    # in front of the optimizer step (joined by the optimizer's gather like every side-stream launch)
                    f._flush_deferred(side=True)
                if seg.bwd_pairs is not None:
                    cur, st = hip._stream(), hip.side_stream(seg.device)
                    for gm, gs in seg.bwd_pairs:
                        gm.replay()
                        if gs is not None:
                            hip.stream_wait(cur, st.cuda_stream)
                            with torch.cuda.stream(st.stream):
                                gs.replay()
                    hip.stream_wait(st.cuda_stream, cur)
                seg.bwd_graph.replay()
                f = seg.fused
                if f.overlap and not f.accumulating:            # bucket bookkeeping of the overlapped all-reduce: this segment's
                    for i in seg.param_index:                   # gradients are in the flat buffer - its buckets' all-reduces start
                        f.param_ready(i)                        # here, under the backward replay of the segment in front
                if seg.static_gin is None:
                    return (None,) * (1 + len(seg.static_in))
                return (None,) + tuple(None if g is None else g.detach() for g in seg.static_gin)

        self._fn = _Replay

    def __call__(self, *inputs):
        return self._fn.apply(self._token, *inputs)

    def release(self):
        """Drops the graphs and every buffer they pinned.  The segment and its `_Replay` closure reference each other, and
        `gc.freeze()` may have moved them to the permanent generation: an evicted segment would otherwise keep its graph
        memory pool for the life of the process."""
        self._fn = None
        self.fwd_graph = self.bwd_graph = self.bwd_pairs = None
        self.static_in = self.static_out = self.static_gout = self._static_grads = self.static_gin = None
        self._keep = []
        self._aliases = []


class _SegmentChain:
    """Graphed segments applied one after the other (backbone, encoder)."""

    def __init__(self, segments):
        self.segments = segments

    def __call__(self, *inputs):
        for seg in self.segments:
            inputs = seg(*inputs)
        return inputs

    def release(self):
        for seg in self.segments:
            seg.release()


class TrainStep:
    """fwd (autocast) -> criterion (fp32) -> bwd -> clip -> AdamW -> [scheduler] -> EMA.

    `hip_graph=True` (GPU, fused optimizer): backbone + encoder - static shapes, ~70 % of the step's kernel launches
    (conv / BN / depthwise units, forward and backward) - are captured once per input shape into HIP graphs
    (GraphedSegment) and replayed, which removes their host-side launch cost (the step was host-paced); the
    decoder (query count depends on the batch's targets) and the criterion (one D2H copy of the
    assignment) stay eager.  A step instrumented with per-launch HIP events (bench.py's roofline sample) runs eagerly."""

    def __init__(self, model, criterion, optimizer, *, amp_dtype=None, clip_max_norm=0.1,
                 ema=None, scheduler=None, accum_steps=1, fused_optimizer=None, hip_graph=False,
                 graph_after=0):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.amp_dtype, self.clip_max_norm = amp_dtype, clip_max_norm
        self.ema, self.scheduler, self.accum_steps = ema, scheduler, max(accum_steps, 1)
        self.iters = 0
        self._micro = 0
        self._params = [p for p in model.parameters() if p.requires_grad]
        self.fused = fused_optimizer      # FusedAdamWEMA (GPU): clip + AdamW + EMA + zero_grad + all-reduce
        self.hip_graph, self.graph_after = hip_graph, graph_after
        self._graphs, self._calls = {}, 0
        self.gc_freeze_after = int(os.environ.get("DFINE_GC_FREEZE_AFTER", "3"))      # 0 = never

    def optimizer_step(self, step_scheduler=True):
        from .. import kernels
        kernels.flush_bn_counters()
        if self.fused is not None:
            self.fused.step()
            if self.hip_graph and self._graphs:
                # the packed / bf16 weight copies the captured forward reads: refreshed right behind the optimizer's kernels
                # instead of at the start of the next step (the device would wait ~0.1 ms for the host to get there)
                kernels.refresh_weight_shadows(self.fused.flat_param.device)
            if step_scheduler and self.scheduler is not None:
                self.scheduler.step()
            self.iters += 1
            return
        if self.clip_max_norm:
            torch.nn.utils.clip_grad_norm_(self._params, self.clip_max_norm, foreach=True)
        self.optimizer.step()
        from .. import kernels
        kernels.bump_weight_epoch()
        if step_scheduler and self.scheduler is not None:
            self.scheduler.step()
        self.optimizer.zero_grad(set_to_none=True)
        if self.ema is not None:
            self.iters += 1
            self.ema.update(self.iters, self.model)

    def _forward(self, images, targets):
        model = self.model
        use_graph = (self.hip_graph and images.is_cuda and self.fused is not None and not isinstance(model, DDP)
                     and hasattr(model, "backbone") and self._calls > self.graph_after and torch.is_grad_enabled())
        if use_graph:
            from .. import hip
            use_graph = not (hip._TIMING_ON or hip._FORCE_EAGER)       # per-launch HIP events only exist for eager launches
        if not use_graph:
            return model(images, targets=targets)
        key = (tuple(images.shape), images.dtype)
        seg = self._graphs.pop(key, None)
        if seg is not None:
            self._graphs[key] = seg              # most recently used last: the eviction below is true LRU
        else:
            # multiscale training draws from 5 input sizes (base, +-32, +-64: reference dataset.py:667-694)
            if len(self._graphs) >= int(os.environ.get("DFINE_GRAPH_SHAPES", "5")):
                self._graphs.pop(next(iter(self._graphs))).release()
            # (single rank: one segment - the split costs ~0.1 ms per step, 31.52 vs 31.38 ms, and buys nothing without an all-reduce)
            split = os.environ.get("DFINE_GRAPH_SPLIT")
            # the images go into the static input as bf16 (the copy per step IS the stem's input cast): HGNetv2's stem is the
            # only reader and takes bf16 under autocast (kernels.conv_bn_act route 4)
            cast_in = self.amp_dtype == torch.bfloat16 and type(model.backbone).__name__ == "HGNetv2"
            if (split == "1") if split is not None else bool(self.fused.overlap):
                # TWO segments, encoder | backbone: the encoder's gradients are delivered to the flat buffer (and its buckets'
                # all-reduces started, data-parallel runs) when ITS backward replay ends, i.e. under the backbone's backward -
                # one segment delivered 55 of the 78 MB of D-FINE-m only after the whole backward.  The encoder reads the
                # backbone's static outputs in place and writes their gradients where the backbone's backward graph reads
                # them: no map is copied between the two.
                bb = GraphedSegment(_Backbone(model.backbone), (images,), amp_dtype=self.amp_dtype, fused=self.fused,
                                    defer_backward=True, cast_inputs=cast_in)
                enc = GraphedSegment(_Encoder(model.encoder), tuple(bb.static_out), amp_dtype=self.amp_dtype, fused=self.fused,
                                     input_grads=True, clone_inputs=False)
                bb.capture_backward(static_gout=enc.static_gin)
                enc.done_before, bb.done_before = ("decoder",), ("decoder", "encoder")
                seg = _SegmentChain([bb, enc])
            else:
                seg = GraphedSegment(_BackboneEncoder(model.backbone, model.encoder), (images,), amp_dtype=self.amp_dtype,
                                     fused=self.fused, cast_inputs=cast_in)
                seg.done_before = ("decoder",)
            self._graphs[key] = seg
        with torch.autocast("cuda", enabled=False):
            feats = seg(images)
        return model.decoder(list(feats), targets)

    def __call__(self, images, targets):
        dev_type = images.device.type
        self._calls += 1
        if images.is_cuda:
            from .. import kernels
            kernels.defer_bn_counters(True)      # one multi-tensor add per step instead of 133 scalar adds
        if self.amp_dtype is not None:
            with torch.autocast(dev_type, dtype=self.amp_dtype):
                outputs = self._forward(images, targets)
        else:
            outputs = self._forward(images, targets)
        with torch.autocast(dev_type, enabled=False):
            loss_dict = self.criterion(outputs, targets)
        total = self.criterion.total(loss_dict) if hasattr(self.criterion, "total") else sum(loss_dict.values())
        loss = total / self.accum_steps
        if self.fused is not None:           # gradient buckets are reduced from backward hooks on the LAST micro-step only
            self.fused.accumulating = (self._micro + 1) % self.accum_steps != 0
        loss.backward()
        self._micro += 1
        if self._micro % self.accum_steps == 0:
            self.optimizer_step()
        if self._calls == self.gc_freeze_after:
            # The model, the optimizer state, the cached launch tables ... are a few hundred thousand long-lived Python
            # objects: every full (generation-2) garbage collection walks all of them - 50-150 ms of host pause, about
            # once per 30-40 steps, during which the device queue of a ~50 ms step runs dry (measured: 1 step in ~35 took
            # 145-210 ms, +3 ms on the mean).  Moving them to the permanent generation once the steady state is reached
            # keeps later collections to the per-step garbage.
            import gc
            gc.collect()
            gc.freeze()
        return loss.detach(), loss_dict
