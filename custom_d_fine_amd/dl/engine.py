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


class GraphedSegment:
    """A module with static shapes whose forward and backward are each captured ONCE into a HIP graph and
    replayed afterwards (two launches instead of thousands of host-side kernel launches).

    Hand-rolled equivalent of torch.cuda.make_graphed_callables - which segfaults on torch 2.10.0+rocm7.0
    even for a two-layer MLP (tools/graph_probe2.py), while explicit torch.cuda.graph capture with
    torch.autograd.grad inside works (tools/graph_probe3.py).  Inputs are copied into static buffers,
    outputs / parameter gradients are returned as the graph's static tensors."""

    def __init__(self, module, sample_inputs, amp_dtype=None, warmup=2):
        from .. import kernels
        kernels._CAPTURE_POSSIBLE = True
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.amp_dtype = amp_dtype
        self.static_in = [t.detach().clone().requires_grad_(t.requires_grad) for t in sample_inputs]
        self.grad_in_idx = [i for i, t in enumerate(self.static_in) if t.requires_grad]

        def run():
            with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None, cache_enabled=False):
                return tuple(module(*self.static_in))

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                outs = run()
                torch.autograd.grad(outs, [self.static_in[i] for i in self.grad_in_idx] + self.params,
                                    [torch.ones_like(o) for o in outs], allow_unused=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            self.static_out = run()
        self.static_gout = [torch.zeros_like(o) for o in self.static_out]
        with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool()):
            self.static_grads = torch.autograd.grad(
                self.static_out, [self.static_in[i] for i in self.grad_in_idx] + self.params,
                self.static_gout, allow_unused=True)
        seg = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *args):           # args = inputs + params (params only to hook autograd up)
                for dst, src in zip(seg.static_in, args[:len(seg.static_in)]):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                seg.fwd_graph.replay()
                return tuple(o.detach() for o in seg.static_out)

            @staticmethod
            def backward(ctx, *gouts):
                for dst, src in zip(seg.static_gout, gouts):
                    if src is None:
                        dst.zero_()
                    elif dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                seg.bwd_graph.replay()
                grads = [None] * len(seg.static_in)
                it = iter(seg.static_grads)
                for i in seg.grad_in_idx:
                    grads[i] = next(it)
                return tuple(grads) + tuple(g.detach() if g is not None else None for g in it)

        self._fn = _Replay

    def __call__(self, *inputs):
        return self._fn.apply(*inputs, *self.params)


class TrainStep:
    """fwd (autocast) -> criterion (fp32) -> bwd -> clip -> AdamW -> [scheduler] -> EMA.

    `hip_graph=True` (GPU): backbone + encoder - static shapes, ~70 % of the step's kernel launches
    (conv / BN / depthwise units, forward and backward) - are captured once into HIP graphs
    and replayed, which removes their host-side launch cost; the
    decoder (query count depends on the batch's targets) and the criterion (one D2H copy of the
    assignment) stay eager.
    Measured on MI355X / ROCm 7.2 (D-FINE-m, bs 32, phases synchronised): forward 25.6 -> 24.5 ms but
    backward 52.5 -> 57.9 ms, i.e. no net gain: the step is bound by device-side kernel boundaries
    (~1.5 us x ~6000 launches, the same for eager and graph launches on this stack), not by host launch
    cost - so the default is OFF and the way forward is fewer, fatter kernels."""

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
        self._graphed, self._calls, self._graph_shape = None, 0, None
        self.gc_freeze_after = int(os.environ.get("DFINE_GC_FREEZE_AFTER", "3"))      # 0 = never

    def optimizer_step(self, step_scheduler=True):
        from .. import kernels
        kernels.flush_bn_counters()
        if self.fused is not None:
            self.fused.step()
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
        use_graph = (self.hip_graph and images.is_cuda and not isinstance(model, DDP)
                     and hasattr(model, "backbone") and self._calls >= self.graph_after)
        if not use_graph:
            return model(images, targets=targets)
        if self._graphed is None or self._graph_shape != tuple(images.shape):
            # Capture on the FIRST call, before any eager .backward(): an AccumulateGrad node created on
            # the default stream by an earlier eager backward makes the later capture segfault on this
            # torch/ROCm build (tools/graph_probe6.py).  GraphedSegment's own warm-up iterations (side
            # stream, torch.autograd.grad) run the conv autotuner / MIOpen find eagerly first.
            be = _BackboneEncoder(model.backbone, model.encoder)
            self._graphed = GraphedSegment(be, (images,), amp_dtype=self.amp_dtype)
            self._graph_shape = tuple(images.shape)
        with torch.autocast("cuda", enabled=False):
            feats = self._graphed(images)
        return model.decoder(list(feats), targets)

    def __call__(self, images, targets):
        dev_type = images.device.type
        self._calls += 1
        if images.is_cuda:
            from .. import kernels
            kernels.defer_bn_counters(True)      # one multi-tensor add per step instead of 133 scalar adds
        if self.amp_dtype is not None:
            with torch.autocast(dev_type, dtype=self.amp_dtype, cache_enabled=not self.hip_graph):
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
