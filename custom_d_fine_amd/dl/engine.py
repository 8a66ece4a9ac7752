"""One optimisation step of the hot path and its satellites (EMA, data-parallel wrap).

Restates the step semantics of the reference's `Trainer.train` / `optimizer_step`
(`src/dl/train.py:512-535,550-586`): autocast forward, fp32 loss, `sum(loss_dict)/accum`,
backward, [unscale ->] clip_grad_norm_(max_norm) -> AdamW -> scheduler -> zero_grad -> EMA.
bf16 autocast needs no GradScaler (the reference's fp16 path does).
"""
import math
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


class TrainStep:
    """fwd (autocast) -> criterion (fp32) -> bwd -> clip -> AdamW -> [scheduler] -> EMA.

    `hip_graph=True` (GPU): backbone + encoder - static shapes, ~70 % of the step's kernel launches
    (conv / BN / depthwise units, forward and backward) - are captured once into HIP graphs
    (torch.cuda.make_graphed_callables) and replayed, which removes their host-side launch cost; the
    decoder (query count depends on the batch's targets) and the criterion (one D2H copy of the
    assignment) stay eager."""

    def __init__(self, model, criterion, optimizer, *, amp_dtype=None, clip_max_norm=0.1,
                 ema=None, scheduler=None, accum_steps=1, fused_optimizer=None, hip_graph=False,
                 graph_after=2):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.amp_dtype, self.clip_max_norm = amp_dtype, clip_max_norm
        self.ema, self.scheduler, self.accum_steps = ema, scheduler, max(accum_steps, 1)
        self.iters = 0
        self._micro = 0
        self._params = [p for p in model.parameters() if p.requires_grad]
        self.fused = fused_optimizer      # FusedAdamWEMA (GPU): clip + AdamW + EMA + zero_grad + all-reduce
        self.hip_graph, self.graph_after = hip_graph, graph_after
        self._graphed, self._calls, self._graph_shape = None, 0, None

    def optimizer_step(self, step_scheduler=True):
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
            # eager warm-up calls (autotuning, MIOpen find, BN buffers) are done: capture fwd + bwd
            be = _BackboneEncoder(model.backbone, model.encoder)
            sample = images.detach().clone()
            self._graphed = torch.cuda.make_graphed_callables(be, (sample,), num_warmup_iters=2)
            self._graph_shape = tuple(images.shape)
        return model.decoder(list(self._graphed(images)), targets)

    def __call__(self, images, targets):
        dev_type = images.device.type
        self._calls += 1
        if self.amp_dtype is not None:
            with torch.autocast(dev_type, dtype=self.amp_dtype, cache_enabled=not self.hip_graph):
                outputs = self._forward(images, targets)
        else:
            outputs = self._forward(images, targets)
        with torch.autocast(dev_type, enabled=False):
            loss_dict = self.criterion(outputs, targets)
        total = self.criterion.total(loss_dict) if hasattr(self.criterion, "total") else sum(loss_dict.values())
        loss = total / self.accum_steps
        loss.backward()
        self._micro += 1
        if self._micro % self.accum_steps == 0:
            self.optimizer_step()
        return loss.detach(), loss_dict
