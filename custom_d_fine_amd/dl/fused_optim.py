"""Flat-buffer optimizer step for the GPU: global-norm clip + AdamW + EMA + zero_grad in
(1 + number of parameter groups + 1) HIP launches (csrc/optim.hip), and data-parallel gradient
averaging as a few large RCCL all-reduces over contiguous ranges of the flat gradient buffer, launched from
backward hooks as soon as a range is complete so that they overlap the rest of the backward pass
(reference: DDP's bucketed reducer, src/dl/train.py:167-179).

The step semantics are the reference's (`src/dl/train.py:512-535`, `ModelEMA` `:52-73`,
parameter groups from `build_optimizer`, `src/d_fine/dfine.py:87-124`); hyper-parameters - including
the per-group learning rates a scheduler may have updated - are read from the torch optimizer that
`build_optimizer` returned, every step.
"""
import math
import os

import torch
import torch.distributed as dist

from ..d_fine.dist_utils import get_world_size, is_dist_available_and_initialized


def _flatten_into(tensors, flat):
    """Copies `tensors` into consecutive slices of `flat` and returns views shaped like them."""
    views, off = [], 0
    for t in tensors:
        n = t.numel()
        v = flat[off:off + n].view_as(t)
        v.copy_(t)
        views.append(v)
        off += n
    return views


class FusedAdamWEMA:
    """Wraps a torch AdamW (for its param_groups / scheduler interface) and a ModelEMA.

    After construction every trainable parameter of `model` (and of the EMA copy) is a view into
    one flat fp32 buffer, gradients are gathered into a flat gradient buffer right before the step,
    and float buffers (BatchNorm statistics) are views into a flat buffer
    too, so EMA is a handful of streaming kernels instead of two launches per tensor.
    """

    def __init__(self, model, optimizer, ema=None, clip_max_norm=0.1, overlap=None, bucket_mb=40):
        """overlap: all-reduce gradient buckets during backward (default: on when world_size > 1; the
        DFINE_GRAD_OVERLAP environment variable overrides).  bucket_mb: xGMI is point-to-point (7 links x ~153 GB/s), a ring
        all-reduce is per-link bound and its fixed latency is paid per call, so buckets are few and large - but only the
        LAST bucket (the first layers of the backbone, complete when backward ends) is exposed, so not too large either.
        Every bucket boundary also joins the side stream and runs the bucket's deferred reductions in the middle of backward:
        measured on a one-rank RCCL group (tools/probe/ddp_mode_timing.py) the mode itself costs +1.06 ms per step with
        16 MB buckets (8 for D-FINE-m's 78 MB), +0.33 ms with 40 MB (5).  DFINE_BUCKET_MB overrides."""
        import os
        from .. import hip
        self.hip = hip
        self.model, self.optimizer, self.ema = model, optimizer, ema
        self.clip_max_norm = float(clip_max_norm or 0.0)
        self.step_count = 0
        self.ema_iters = 0
        self._live = []
        dev = next(model.parameters()).device
        assert dev.type == "cuda", "FusedAdamWEMA drives HIP kernels; parameters must live on the GPU"

        groups = [[p for p in g["params"] if p.requires_grad] for g in optimizer.param_groups]
        for g in groups:
            for p in g:
                assert p.dtype == torch.float32
        sizes = [sum(p.numel() for p in g) for g in groups]
        # keep each group 16-byte aligned for the float4 kernels
        padded = [(s + 3) // 4 * 4 for s in sizes]
        total = sum(padded)
        self.flat_param = torch.zeros(total, device=dev)
        self.flat_grad = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        self.flat_ema = torch.zeros(total, device=dev) if ema is not None else None
        self.sqnorm = self.hip.grad_sqnorm_buffer(dev)      # [0] = squared norm, rest scratch
        self.segments = []
        self._params, self._grad_views, self._grad_offsets = [], [], []
        ema_params = dict(ema.model.named_parameters()) if ema is not None else {}
        names = {id(p): n for n, p in model.named_parameters()}
        off = 0
        for g, size, pad in zip(groups, sizes, padded):
            seg = slice(off, off + size)
            views = _flatten_into([p.data for p in g], self.flat_param[seg])
            gviews = _flatten_into([torch.zeros_like(p) for p in g], self.flat_grad[seg])
            for p, v, gv in zip(g, views, gviews):
                p.data = v
            self._params.extend(g)
            self._grad_views.extend(gviews)
            o = off
            for p in g:
                self._grad_offsets.append(o)
                o += p.numel()
            if ema is not None:
                eps_ = [ema_params[names[id(p)].replace("module.", "", 1) if names[id(p)].startswith("module.")
                                   else names[id(p)]] for p in g]
                eviews = _flatten_into([e.data for e in eps_], self.flat_ema[seg])
                for e, v in zip(eps_, eviews):
                    e.data = v
            self.segments.append((off, size))
            off += pad

        # float buffers (BatchNorm running statistics, anchors, ...) for the EMA of the buffers
        self.flat_buf = self.flat_ema_buf = None
        if ema is not None:
            stu_bufs = [(n, b) for n, b in model.named_buffers() if b.dtype == torch.float32]
            ema_bufs = dict(ema.model.named_buffers())
            if stu_bufs:
                nb = sum(b.numel() for _, b in stu_bufs)
                self.flat_buf = torch.zeros(nb, device=dev)
                self.flat_ema_buf = torch.zeros(nb, device=dev)
                sviews = _flatten_into([b for _, b in stu_bufs], self.flat_buf)
                eviews = _flatten_into([ema_bufs[n.replace("module.", "", 1) if n.startswith("module.") else n]
                                        for n, _ in stu_bufs], self.flat_ema_buf)
                for (n, b), sv, ev in zip(stu_bufs, sviews, eviews):
                    b.data = sv                # in-place kernels (BN running stats) now write the flat buffer
                    key = n.replace("module.", "", 1) if n.startswith("module.") else n
                    ema_bufs[key].data = ev
            # parameters outside the optimizer (frozen) never change: their EMA stays equal

        # ---- gradient buckets: contiguous flat ranges, filled from the END of every parameter group (backward produces
        # the gradients roughly in reverse forward order); a bucket is reduced as soon as its last gradient arrived
        env = os.environ.get("DFINE_GRAD_OVERLAP")
        self.overlap = (get_world_size() > 1) if overlap is None else bool(overlap)
        if env is not None:
            self.overlap = env == "1"
        self.accumulating = False            # TrainStep sets it on all but the last micro-step of an accumulation window
        self._buckets, self._works = [], []
        bucket_mb = float(os.environ.get("DFINE_BUCKET_MB", bucket_mb))
        cap = max(int(bucket_mb * (1 << 20) // 4), 1)
        pidx = 0
        index_of = {}

        def top_module(p):
            n = names[id(p)]
            n = n[7:] if n.startswith("module.") else n
            return n.split(".", 1)[0]

        for g, (off, size) in zip(groups, self.segments):
            entries = []
            for p in g:
                entries.append((pidx, self._grad_offsets[pidx], p.numel(), top_module(p)))
                pidx += 1
            cur, hi = [], off + size
            rev = list(reversed(entries))
            for j, e in enumerate(rev):
                cur.append(e)
                # a bucket also ends where the top-level module changes (decoder | encoder | backbone): the decoder's
                # gradients are complete when ITS backward ends, and their all-reduce then overlaps the whole backward of
                # encoder + backbone (15 ms of a 33 ms step) instead of waiting for the encoder weights of a mixed bucket -
                # with the captured backward segment (dl/engine.GraphedSegment) those only arrive when its last graph has run
                boundary = j + 1 < len(rev) and rev[j + 1][3] != e[3]
                if hi - e[1] >= cap or boundary:
                    self._buckets.append({"lo": e[1], "hi": hi, "params": [c[0] for c in cur], "ready": 0, "done": False,
                                          "owner": e[3]})
                    cur, hi = [], e[1]
            if cur:
                self._buckets.append({"lo": cur[-1][1], "hi": hi, "params": [c[0] for c in cur], "ready": 0, "done": False,
                                      "owner": cur[-1][3]})
        # Collectives pair up across ranks by CALL ORDER, so the all-reduces are issued in one fixed order on every rank - the
        # order backward is expected to complete the buckets in (latest-registered parameters first) - and a bucket whose
        # turn has not come waits, gathered, for its predecessors (DDP's reducer does the same; a rank-local completion order
        # would pair different ranges on different ranks when one rank has a parameter without gradient, e.g. an image
        # batch without targets skips the denoising embedding).
        # The order: by top-level module in the order their BACKWARD passes end (decoder, encoder, backbone - the reverse of the
        # forward; `named_parameters()` lists them in attribute order, which says nothing about execution), inside a module the
        # latest-registered parameters first.  (Until round 5 the key was the parameter index alone: the encoder's buckets came
        # first and the decoder's - complete long before - waited behind them.)
        order = {m: i for i, m in enumerate(getattr(model, "_dfine_backward_order", ("decoder", "encoder", "backbone")))}
        self._buckets.sort(key=lambda b: (order.get(b["owner"], -1), -max(b["params"])))
        self._next_launch = 0
        for bi, b in enumerate(self._buckets):
            b["gathered"] = False
            for i in b["params"]:
                index_of[i] = bi
        if self.overlap:
            for i, p in enumerate(self._params):
                p.register_post_accumulate_grad_hook(self._make_hook(index_of[i]))
        # ---- deferred weight gradients: the conv / linear backward ops leave their split partial sums in a workspace and
        # register them here; ONE launch per flush reduces all of them straight into the flat gradient buffer
        self._bucket_of = index_of
        self._deferred = []          # (param index, dst offset in flat_grad, workspace tensor, workspace offset, meta)
        self._uses = {}              # param index -> deferred uses still to come in this backward
        self.defer_wgrads = True
        for i, p in enumerate(self._params):
            p._dfine_slot = (self, i)

    # -------------------------------------------------------------------------------------
    def broadcast_from_rank0(self):
        """Every rank starts from rank 0's model, like the reference's DDP wrap (train.py:167-179 broadcasts the whole
        module state): trainable parameters and float buffers through the flat buffers, then whatever lives outside them
        (frozen parameters - the l/x configs freeze the HGNetv2 stem - and integer buffers), then the EMA copy is re-derived
        from the synchronised student."""
        if not (is_dist_available_and_initialized() and get_world_size() > 1):
            return
        dist.broadcast(self.flat_param, 0)
        if self.flat_buf is not None:
            dist.broadcast(self.flat_buf, 0)
        flat_ids = {id(p) for p in self._params}
        rest = [p.data for p in self.model.parameters() if id(p) not in flat_ids]
        rest += [b.data for b in self.model.buffers() if self.flat_buf is None or b.dtype != torch.float32]
        for t in rest:
            if t.numel():
                dist.broadcast(t, 0)
        from ..d_fine.arch.utils import invalidate_weighting_cache
        invalidate_weighting_cache()         # `up` / `reg_scale` are frozen parameters: just rewritten through .data
        from .. import kernels
        kernels.bump_weight_epoch()          # ... and the cached packed / bf16 weight copies are stale
        if self.ema is not None:
            self.flat_ema.copy_(self.flat_param)
            if self.flat_ema_buf is not None:
                self.flat_ema_buf.copy_(self.flat_buf)
            stu = dict(self.model.named_parameters())
            stu.update(dict(self.model.named_buffers()))
            with torch.no_grad():
                for n, t in list(self.ema.model.named_parameters()) + list(self.ema.model.named_buffers()):
                    src = stu.get(n, stu.get("module." + n))
                    if src is not None and t.data_ptr() != src.data_ptr() and not (
                            self.flat_ema.data_ptr() <= t.data_ptr() < self.flat_ema.data_ptr() + self.flat_ema.numel() * 4):
                        t.copy_(src)

    def state_dict(self):
        """What a resumed run needs beyond the model / EMA state dicts: the Adam moments (they live only in the flat
        buffers, `optimizer.state_dict()` stays empty) and the two step counters.  Layout = the flat parameter layout of
        this model and parameter-group order, checked on load."""
        return {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                "step_count": int(self.step_count), "ema_iters": int(self.ema_iters),
                "segments": [tuple(s) for s in self.segments]}

    def load_state_dict(self, state):
        if [tuple(s) for s in state["segments"]] != [tuple(s) for s in self.segments]:
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        self.step_count = int(state["step_count"])
        self.ema_iters = int(state["ema_iters"])

    def zero_grad(self):
        self.flat_grad.zero_()
        for p in self._params:
            p.grad = None

    def defer_wgrad(self, index, ws, meta, ws_offset=0, dst_offset=0):
        """Called from a backward op instead of returning a gradient tensor for parameter `index`: `ws` holds per-split
        partial sums (layout `meta` = (splits, Cout, Cin, taps, NP16, CP16)) of the gradient of the parameter's elements
        [dst_offset, dst_offset + Cout * Cin * taps)."""
        # (measured and dropped: an early reduction of every n registered entries on the side stream - no gain on the device, the
        # side stream is the one that finishes last, and ~0.2 ms of host time per flush)
        self._deferred.append((index, self._grad_offsets[index] + dst_offset, ws, ws_offset, meta))

    def grad_offset(self, index):
        return self._grad_offsets[index]

    def grad_ptr(self, index):
        """Device address of parameter `index`'s slot in the flat gradient buffer (for kernels that ADD into it)."""
        return self.flat_grad.data_ptr() + 4 * self._grad_offsets[index]

    def note_use(self, index):
        """A forward op will deliver parameter `index`'s gradient through defer_wgrad (one call per use of the parameter)."""
        self._uses[index] = self._uses.get(index, 0) + 1

    def use_done(self, index):
        n = self._uses.get(index, 0) - 1
        self._uses[index] = n
        if n <= 0:
            self.param_ready(index)

    def param_ready(self, index):
        """The gradient of parameter `index` is complete (all its deferred pieces registered): bucket bookkeeping of the
        overlapped all-reduce - what the post-accumulate-grad hook does for parameters that get a gradient tensor."""
        if self.overlap and not self.accumulating:
            b = self._buckets[self._bucket_of[index]]
            if b["gathered"]:
                raise RuntimeError("a gradient was registered for a bucket that was already handed to the all-reduce")
            b["ready"] += 1
            if b["ready"] == len(b["params"]):
                self._reduce_bucket(b)

    def module_backward_done(self, owners):
        """The backward passes of the top-level modules `owners` are over: their buckets that are still incomplete - a parameter
        that gets NO gradient in this step (an unused head, the denoising embedding of a batch without targets) never reports -
        are handed to the all-reduce as they are (missing gradients are zeros in the flat buffer).  Without this, one such
        parameter keeps its bucket, and every bucket behind it in the fixed launch order, until the end of backward.  Called by
        the graphed segments (dl/engine.py) at the start of their backward replay: deterministic, the same on every rank."""
        if not self.overlap or self.accumulating:
            return
        for b in self._buckets:
            if b["owner"] in owners and not b["gathered"]:
                self._reduce_bucket(b)

    def _reduce_blocks(self, splits, elems, _memo={}):
        key = (splits, elems)
        if key not in _memo:
            _memo[key] = self.hip.multi_wgrad_reduce_blocks(splits, elems)
        return _memo[key]

    def _flush_deferred(self, lo=None, hi=None, side=False):
        """side: grouped launches and reduction on the side stream (hip._side_fork), nothing joined - a later flush does."""
        import numpy as np
        # registered weight-gradient GEMMs -> their partial sums (one launch per kind); joins the side stream unless side
        self.hip.linear_wgrad_flush(side=side)
        take = [d for d in self._deferred if lo is None or lo <= d[1] < hi]
        if not take:
            return
        if lo is not None:
            self._deferred = [d for d in self._deferred if not (lo <= d[1] < hi)]
        else:
            self._deferred = []
        fg = self.flat_grad.data_ptr()
        rows = [(ws.data_ptr() + 4 * wo, fg + 4 * off, m[0], m[1], m[2], m[3], m[4], m[5]) for _, off, ws, wo, m in take]
        from ..d_fine.arch.utils import upload
        # The kernel adds one table row per blockIdx.y into its destination with a plain read-modify-write: rows that share a
        # destination (a module applied several times in one forward - query_pos_head runs once per decoder layer - or one
        # row per micro-step of a gradient-accumulation window) must not run in the same launch.  Round r holds the r-th row
        # of every destination; the rounds are stream-ordered launches, so the sum order is fixed (replicas stay bit-identical).
        seen, rounds = {}, []
        for row in rows:
            r = seen.get(row[1], 0)
            seen[row[1]] = r + 1
            if r == len(rounds):
                rounds.append([])
            rounds[r].append(row)
        order = [row for rnd in rounds for row in rnd]
        table = upload(np.asarray(order, dtype=np.int64), self.flat_grad.device)
        first = 0
        for rnd in rounds:
            blocks = max(self._reduce_blocks(r[2], r[3] * r[4] * r[5]) for r in rnd)
            io = sum(4.0 * r[2] * r[6] * r[7] * r[5] + 8.0 * r[3] * r[4] * r[5] for r in rnd)
            self.hip.multi_wgrad_reduce(table[first:first + len(rnd)], len(rnd), blocks, io=io, side=side)
            first += len(rnd)
        self._live.append((take, table))

    def _make_hook(self, bi):
        b = self._buckets[bi]
        n = len(b["params"])

        def hook(param):
            # (the hook also fires, with .grad still None, for a parameter whose backward op returned no tensor because it
            # delivered the gradient through defer_wgrad / straight into the flat buffer: those report through param_ready)
            if self.accumulating or param.grad is None:
                return
            if b["gathered"]:
                raise RuntimeError("a gradient arrived for a bucket that was already handed to the all-reduce "
                                   "(module_backward_done called too early?)")
            b["ready"] += 1
            if b["ready"] == n:
                self._reduce_bucket(b)
        return hook

    def _gather(self, indices):
        """Moves the listed parameters' gradients into their flat slots with one multi-tensor copy and drops them.
        (Keeping `.grad` as persistent views instead makes autograd ADD into them - one tiny kernel per parameter,
        646 per step for D-FINE-m.)"""
        self.hip.side_join()                 # gradient tensors produced on the side stream (depthwise / stem weight gradients)
        grads, offs = [], []
        for i in indices:
            p = self._params[i]
            g = p.grad
            if g is not None:
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                grads.append(g)
                offs.append(self._grad_offsets[i])
                p.grad = None
        if grads:
            # one HIP launch driven by a pointer table (torch._foreach_copy_ degrades to one DtoD memcpy
            # per tensor here: 646 launches per step for D-FINE-m)
            self._live.append((grads, self.hip.multi_copy_f32(grads, offs, self.flat_grad)))

    def _reduce_bucket(self, b):
        self._gather(b["params"])
        self._flush_deferred(b["lo"], b["hi"])
        b["gathered"] = True
        self._launch_in_order()

    def _launch_in_order(self, force=False):
        """Starts the all-reduce of every bucket whose turn has come (all earlier buckets of the fixed order launched)."""
        while self._next_launch < len(self._buckets):
            b = self._buckets[self._next_launch]
            if not b["gathered"]:
                if not force:
                    return
                self._gather(b["params"])
                self._flush_deferred(b["lo"], b["hi"])
                b["gathered"] = True
            b["done"] = True
            self._next_launch += 1
            if get_world_size() > 1:
                # asynchronous on the communication stream: ordered after the copies above, overlaps what backward still runs
                self._works.append(dist.all_reduce(self.flat_grad[b["lo"]:b["hi"]], async_op=True))

    def _collect_grads(self):
        """Single-shot path (no hooks) and flush of whatever the hooks did not see (parameters without a gradient this
        step leave their bucket incomplete)."""
        if not self.overlap:
            self._gather(range(len(self._params)))
            self._flush_deferred()
            if get_world_size() > 1:
                # one large collective over xGMI: 78 MB for D-FINE-m
                dist.all_reduce(self.flat_grad)
            return
        self._launch_in_order(force=True)      # parameters without a gradient this step leave their bucket incomplete
        if self._deferred:
            raise RuntimeError(f"{len(self._deferred)} deferred weight gradients were registered after their bucket had "
                               "been reduced (bucket bookkeeping out of step with the backward ops)")
        for w in self._works:
            w.wait()                         # the compute stream waits for the communication stream; the host does not block
        self._works.clear()
        for b in self._buckets:
            b["ready"], b["done"], b["gathered"] = 0, False, False
        self._next_launch = 0

    def step(self):
        """all-reduce (if data parallel) -> norm -> per-group AdamW+EMA (also zeroes the grads)."""
        hip = self.hip
        world = get_world_size()
        self._collect_grads()
        self._uses.clear()
        grad_scale = 1.0 / world
        self.step_count += 1
        if self.clip_max_norm > 0:
            hip.grad_sqnorm(self.flat_grad, grad_scale, self.sqnorm)
        mom = 0.0
        if self.ema is not None:
            self.ema_iters += 1
            mom = self.ema.ema_scheduler(self.ema_iters)
        for (off, size), group in zip(self.segments, self.optimizer.param_groups):
            if size == 0:
                continue
            b1, b2 = group["betas"]
            seg = slice(off, off + size)
            hip.adamw_ema_step(self.flat_param[seg], self.flat_grad[seg], self.exp_avg[seg], self.exp_avg_sq[seg],
                               self.flat_ema[seg] if self.flat_ema is not None else None,
                               self.sqnorm if self.clip_max_norm > 0 else None, group["lr"], b1, b2, group["eps"],
                               group["weight_decay"], self.step_count, grad_scale, self.clip_max_norm, mom)
        if self.flat_buf is not None:
            hip.ema_update(self.flat_ema_buf, self.flat_buf, mom)
        from .. import kernels
        kernels.bump_weight_epoch()          # cached packed conv weights are stale now
        # keep torch's scheduler bookkeeping consistent (it warns if optimizer.step was never called)
        self.optimizer._opt_called = True
        self._live = self._live[-16:]        # sources of the copies / reductions queued this step; older ones have long been consumed
