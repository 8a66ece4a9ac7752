"""Headline benchmark: images/sec of one full D-FINE-m 640x640 train step, bs=32 per GPU
(BASELINE.json metric / configs[2]) - fwd (bf16 autocast) + Hungarian matcher + criterion (fp32)
+ bwd + grad clip + AdamW + EMA on a device-resident synthetic batch.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself through
                                                             torch.distributed.run, one rank per GPU over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  `value` = images of all ranks / wall time of the K timed
steps (barrier + device synchronize on both sides, max over ranks; no instrumentation inside - every sampled / traced step runs
after the second fence); `median_ms_per_step` = median of the K per-step
durations (HIP events around every step).  Extra objects:
  roofline          the dominant kernel FAMILY by device time: the dense-convolution implicit GEMMs of backbone + encoder
                    (forward, data gradient, weight gradient, stem; MFMA roof).  achieved = algorithmic FLOPs of the family's
                    launches (2 B HW Cin Cout KS^2 each, counted by hip.py `_timed` in the instrumented eager step) / the
                    family's kernel time per step IN THE TIMED MODE - graph replay of backbone + encoder with the weight
                    gradients on a second stream -, taken from the device timestamps of every kernel of `--trace-steps` more
                    steps run right after the timed region under torch.profiler's in-process tracer (HIP events cannot
                    bracket kernels launched from a graph replay).  Sub-objects: `events` = the same sum from HIP events
                    around every launch of `--event-steps` eager two-stream steps run AFTER the timed region, `isolated`
                    = one eager step with everything on one stream (no neighbour kernels), `profile` = the figure
                    recomputed from the committed rocprofv3 --kernel-trace --stats summary of this command
                    (tools/roofline_from_stats.py; same kernel-name patterns, KERNEL_GROUPS).
  roofline_kernels  the same figures per kernel group (1x1 / 3x3 forward+dgrad, weight gradients, stem, token-stream
                    linears / attention) and the two deformable-attention kernels against the HBM roof
                    (algorithmic bytes of SURVEY.md 8(d); `traffic` = PMC HBM bytes from profiles/).
  cpu_baseline      the same train step through the CPU oracle backend ("port") on the host cores, D-FINE-m 640x640 at
                    bs=2, bounded to a few steps.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LRS = {"n": (8e-4, 4e-4), "s": (2.5e-4, 6e-5), "m": (1.5e-4, 2e-5), "l": (1.6e-4, 1e-5), "x": (2e-4, 2e-6)}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFS = 157.3    # f32-input MFMA peak = the fp32 vector rate (MI355X_MICROARCH.md): the roof of --dtype fp32 (config #2)
MFMA_GROUPS = ("conv1x1", "conv3x3", "conv1x1_wgrad", "conv3x3_wgrad", "wgrad_reduce", "stem_conv", "stem_wgrad", "miopen_conv")
F32_GROUPS = ("conv_f32", "linear_f32")       # event keys of the fp32 kernels (convolutions; GEMMs of 1x1 convolutions, linears, attention)


# kernel-name patterns of the groups the line reports (shared with tools/roofline_from_stats.py, which applies them to a
# rocprofv3 --kernel-trace --stats summary of this command)
KERNEL_GROUPS = (
    ("conv1x1", r"dfine::conv1x1_(glds|tr|ring)_kernel"),
    ("conv3x3", r"dfine::(conv_igemm_kernel<3|conv3x3_ws_kernel|conv3x3_rows32_kernel)"),
    ("conv1x1_wgrad", r"dfine::(conv_wgrad1_glds_kernel|conv_wgrad1_group_kernel|conv_wgrad_kernel<1>)"),
    ("conv3x3_wgrad", r"dfine::(conv_wgrad_kernel<3>|conv_wgrad3_)"),
    ("stem", r"dfine::(stem_(conv|mfma|dgrad|wgrad)|stem3_(fwd|bwd)_rows)"),
    ("wgrad_reduce", r"dfine::(multi_wgrad_reduce_kernel|conv_wgrad_reduce_kernel)"),
    ("linear_wgrad", r"dfine::linear_wgrad"),
    ("linear+attention", r"dfine::(linear_(act|ring)_kernel|attn_)"),
    ("msda_fwd", r"dfine::msda_fwd"),
    ("msda_bwd", r"dfine::msda_bwd"),
    ("batchnorm", r"dfine::bn2?_"),
    ("depthwise", r"dfine::dwconv_"),
    ("aten", r"at::native::"),
    ("f32_mfma", r"dfine::(conv_f32_kernel|wgrad_f32_kernel|gemm_f32_(big|nt)_kernel)"),
)
FAMILY_GROUPS = ("conv1x1", "conv3x3", "conv1x1_wgrad", "conv3x3_wgrad", "stem", "wgrad_reduce")
# event keys (hip.py `_timed`) -> traced kernel group
EVENT_TO_GROUP = {"conv1x1": "conv1x1", "conv3x3": "conv3x3", "conv1x1_wgrad": "conv1x1_wgrad", "conv3x3_wgrad": "conv3x3_wgrad",
                  "wgrad_reduce": "wgrad_reduce", "stem_conv": "stem", "stem_wgrad": "stem", "linear_wgrad": "linear_wgrad",
                  "linear": "linear+attention", "attention": "linear+attention", "msda_fwd": "msda_fwd", "msda_bwd": "msda_bwd",
                  "conv_f32": "f32_mfma", "linear_f32": "f32_mfma"}


def traced_groups(step, images, targets, n_steps=3):
    """Kernel time per group and step in the TIMED mode (graph replay of backbone + encoder, weight gradients on the second
    stream): device timestamps of every kernel of `n_steps` more steps, collected in-process through torch.profiler (roctracer
    activity records - kernels launched from graph replays are traced like any other; HIP events cannot bracket them).
    -> {group: (ms per step, launches per step)} + {"*": total kernel ms per step}, or None when the tracer is unavailable."""
    import re
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(n_steps):
                step(images, targets)
            torch.cuda.synchronize()
        rows = [(k.key, k.device_time_total / 1e3, k.count) for k in prof.key_averages() if k.device_time_total > 0]
    except Exception as e:                                   # noqa: BLE001 - the line is still valid without this object
        print(f"torch.profiler trace failed: {e!r}", file=sys.stderr)
        return None
    out = {}
    for name, pat in KERNEL_GROUPS:
        rx = re.compile(pat)
        sel = [r for r in rows if rx.search(r[0])]
        out[name] = (sum(r[1] for r in sel) / n_steps, sum(r[2] for r in sel) / n_steps)
    out["*"] = (sum(r[1] for r in rows) / n_steps, sum(r[2] for r in rows) / n_steps)
    return out


def build_step(model_name, img, device, amp_dtype, num_classes=80, channels_last=False, mask=False):
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep, wrap_data_parallel
    base_lr, backbone_lr = LRS[model_name]
    model = dfine.build_model(model_name, num_classes, mask, str(device), img_size=[img, img]).train()
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    criterion = dfine.build_loss(model_name, num_classes, 0.0, mask)
    ema = ModelEMA(model, 0.9998)
    fused = None
    opt = dfine.build_optimizer(model, lr=base_lr, backbone_lr=backbone_lr, betas=(0.9, 0.999),
                                weight_decay=1.25e-4, base_lr=base_lr)
    if device.type == "cuda" and os.environ.get("DFINE_FUSED_OPT", "1") == "1":
        # flat-buffer clip + AdamW + EMA kernels; data parallelism = bucketed all-reduce of the flat grads
        from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
        fused = FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1)
        fused.broadcast_from_rank0()
    else:
        model = wrap_data_parallel(model, device)
        opt = dfine.build_optimizer(model, lr=base_lr, backbone_lr=backbone_lr, betas=(0.9, 0.999),
                                    weight_decay=1.25e-4, base_lr=base_lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=base_lr * 2, total_steps=100000,
                                                pct_start=0.1, cycle_momentum=False)
    return TrainStep(model, criterion, opt, amp_dtype=amp_dtype, clip_max_norm=0.1, ema=ema,
                     scheduler=sched, fused_optimizer=fused,
                     hip_graph=device.type == "cuda" and fused is not None and os.environ.get("DFINE_HIPGRAPH", "1") == "1")


def msda_algorithmic_bytes(batch, lq, heads=8, head_dim=32, points=12, elt=2, backward=False):
    from custom_d_fine_amd import hip
    return hip.msda_algorithmic_bytes(batch, lq, heads, head_dim, points, elt, backward)


def pmc_traffic(kernel, batch, lq, dtype):
    """HBM bytes per launch from the committed PMC passes (profiles/*_msda_pmc.json: list of records), or None when
    no record matches this run's kernel and shape."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if not name.endswith("_msda_pmc.json"):
            continue
        recs = json.load(open(os.path.join(ROOT, "profiles", name)))
        for rec in recs if isinstance(recs, list) else [recs]:
            sh = rec.get("shape", {})
            if rec.get("bench_key") == kernel and (sh.get("B"), sh.get("Lq"), sh.get("dtype")) == (batch, lq, dtype):
                best = rec["traffic_bytes_per_launch"]           # later rounds override earlier ones
    return best


def conv_pmc_traffic():
    """Measured HBM bytes per launch of the family's reference kernel (conv1x1_glds_kernel on the 512 -> 512 @ 80x80 layer,
    profiles/*_conv_pmc.json: FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) next to its compulsory bytes, or None."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.endswith("_conv_pmc.json"):
            best = json.load(open(os.path.join(ROOT, "profiles", name)))
            if isinstance(best, dict):
                # a committed counter pass of an earlier run on another box, not a measurement of THIS run
                best = dict(best, source=f"profiles/{name}", measured_in_run=False)
    return best


def profile_roofline():
    """The family's fraction recomputed from the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN_roofline_from_profile.json, written by tools/roofline_from_stats.py from rNN_bench_kernel_stats.csv): kernel
    time of the family in the GRAPH-REPLAY two-stream mode that produces `value`.  Not a measurement of this run."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.endswith("_roofline_from_profile.json"):
            best = dict(json.load(open(os.path.join(ROOT, "profiles", name))), source=f"profiles/{name}", measured_in_run=False)
    return best


def cpu_baseline(model_name, img, steps):
    """The same train step on the host through the oracle backend (kind 'port'), as BASELINE.md section 3 asks: fp32, bs 2,
    one intra-op thread per PHYSICAL core, 1 warm-up + `steps` timed steps, median."""
    from oracle import torch_backend
    from custom_d_fine_amd.dl.synthetic import make_batch
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    before = torch.get_num_threads()
    torch.set_num_threads(int(cores))
    torch_backend.install()
    try:
        bs = 2
        step = build_step(model_name, img, torch.device("cpu"), None)
        images, targets = make_batch(bs, img, seed=42)
        step(images, targets)  # warm-up (allocator, thread pools)
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step(images, targets)
            times.append(time.perf_counter() - t0)
    finally:
        torch_backend.uninstall()
        torch.set_num_threads(before)
    med = statistics.median(times)
    return {"value": round(bs / med, 4), "unit": "images/sec", "cores": int(cores),
            "kind": "port",
            "sample": f"D-FINE-{model_name} {img}x{img} bs={bs} fp32 full train step, 1 warm-up + {steps} timed steps (median "
                      f"{med:.2f} s, total {sum(times):.1f} s) through oracle/torch_backend.py, {cores} intra-op threads = physical "
                      f"cores of {os.cpu_count()} logical host CPUs"}


def _self_spawn(n):
    """`python bench.py --gpus N` without a launcher: one rank per GPU through torch.distributed.run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


_GC_LOG = []


def _gc_probe(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    elif info["generation"] >= 1:
        _GC_LOG.append((info["generation"], round((time.perf_counter() - _t[0]) * 1e3, 1)))


def main():
    if os.environ.get("DFINE_BENCH_DUMP_STEPS") == "1":
        import gc
        gc.callbacks.append(_gc_probe)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="m")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mask", type=int, default=0, help="1: segmentation head (BASELINE configs[4])")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (0 = skip)")
    ap.add_argument("--event-steps", type=int, default=1,
                    help="eager steps with HIP events around the kernel launches, run AFTER the timed region (0 = no roofline objects)")
    ap.add_argument("--channels-last", type=int, default=0)
    ap.add_argument("--dry", action="store_true",
                    help="launcher check: spawn the ranks (capped at the visible devices), initialise the RCCL process group, run one "
                         "all-reduce + barrier, print one JSON line and stop (no model, no timing)")
    ap.add_argument("--trace-steps", type=int, default=3,
                    help="steps run AFTER the timed region under torch.profiler's kernel tracer (0 = none): per-kernel device time in "
                         "the timed (graph-replay, two-stream) mode, the source of `roofline.achieved`")
    args = ap.parse_args()

    if args.dry and "WORLD_SIZE" not in os.environ:
        args.gpus = max(1, min(args.gpus, torch.cuda.device_count()))
        sys.argv = [a for a in sys.argv if not a.startswith("--gpus")]          # ("--gpus N" as two tokens: drop the value too)
        sys.argv = [a for i, a in enumerate(sys.argv) if not (i > 0 and sys.argv[i - 1] == "--gpus")] + ["--gpus", str(args.gpus)]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the HIP path and needs an MI355X (no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", device_id=device)

    if args.dry:
        t = torch.ones(1, device=device)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        torch.cuda.synchronize()
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "all_reduce_of_ones": t.item(), "devices_visible": torch.cuda.device_count(),
                              "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.dl.synthetic import make_batch
    torch.manual_seed(42)        # identical initial weights on every rank; the per-rank stream is the DATA (seed 42 + rank)
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    step = build_step(args.model, args.img, device, amp, channels_last=bool(args.channels_last), mask=bool(args.mask))
    images, targets = make_batch(args.batch, args.img, seed=42 + rank, device=device, with_masks=bool(args.mask))
    if args.channels_last:
        images = images.contiguous(memory_format=torch.channels_last)

    # W untimed warm-up steps (at least two: the second builds the batched weight-pack / bf16-shadow tables)
    for _ in range(max(args.warmup, 2)):
        step(images, list(targets))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.freeze()                  # the long-lived heap (model, optimizer, caches) stays out of the cyclic collector's walks
    # events only around the launches the line reports: every event pair costs the host ~5 us, and a sampled step that is
    # host-paced overlaps its two streams less than the un-instrumented steps do
    hip.enable_timing(MFMA_GROUPS + F32_GROUPS + ("linear_wgrad", "linear", "attention", "msda_fwd", "msda_bwd"))
    hip.timing_active(False)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # a NEW target list per step, as a data loader hands over: the per-batch caches of matcher / criterion (target masks at
        # mask resolution, concatenated labels / boxes) are keyed on the list and must be rebuilt every step like in training
        step(images, list(targets))
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    # ---- everything below is instrumentation and runs OUTSIDE the timed region: `value` / `ms_per_step` come from the K
    # un-instrumented steps above only (reference methodology: src/dl/bench.py:80-120 times only the call under test)
    sampled_steps = list(range(args.event_steps))
    eager_ms = []
    if sampled_steps:
        # `--event-steps` eager steps with HIP events around every reported launch (two streams, like the timed mode but
        # host-paced), after one un-instrumented eager step: the FIRST eager step of a process grows the caching allocator's
        # default-stream pool (the graph replays live in their own pool) - `eager_first_step` records what that costs
        for k in range(1 + args.event_steps):
            hip.timing_active(k > 0)
            r0 = torch.cuda.memory_reserved(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            hip.force_eager(True)
            e0.record()
            step(images, list(targets))
            e1.record()
            hip.force_eager(False)
            torch.cuda.synchronize()
            eager_ms.append({"ms": round(e0.elapsed_time(e1), 2), "host_ms": round((time.perf_counter() - h0) * 1e3, 2),
                             "reserved_growth_mb": round((torch.cuda.memory_reserved(device) - r0) / 2**20, 1),
                             "instrumented": k > 0})
        hip.timing_active(False)
        # one more step with everything on one stream: the same kernels without a neighbour (records under "iso:<key>")
        hip.timing_active(True, isolated=True)
        step(images, targets)
        hip.timing_active(False)
        torch.cuda.synchronize()
    traced = None
    if sampled_steps and args.trace_steps > 0:
        # ... and a few more in the timed mode under the in-process kernel tracer (every rank runs the steps, rank 0 traces)
        if rank == 0:
            traced = traced_groups(step, images, targets, args.trace_steps)
        else:
            for _ in range(args.trace_steps):
                step(images, targets)
        torch.cuda.synchronize()
    if rank == 0:
        if os.environ.get("DFINE_BENCH_DUMP_STEPS") == "1":
            print("gc collections (generation, ms):", _GC_LOG, file=sys.stderr)
        print("per-step ms:", " ".join(f"{t:.1f}" for t in per_step), file=sys.stderr)
    timing = hip.timing_summary()
    hip.disable_timing()
    per_rank = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [e.item() for e in every]          # the ranks' own clocks: the line reports the slowest
        elapsed = max(per_rank)

    if rank == 0:
        max_t = max(len(t["labels"]) for t in targets)
        dn = 2 * max_t * max(100 // max_t, 1)
        lq = 300 + dn

        MODES = {"traced": "timed mode: graph replay of backbone + encoder, weight gradients on the second stream; device timestamps of "
                           f"every kernel of {args.trace_steps} steps run right after the timed region, collected in-process through "
                           "torch.profiler (roctracer)",
                 "events": "eager step(s) after the timed region on the same two streams, HIP events on the stream each launch goes to",
                 "isolated": "one eager step after the timed region with every launch on ONE stream (no neighbour kernel), HIP events"}

        def event_sums(keys, iso):
            keys = tuple("iso:" + k for k in keys) if iso else keys
            per = 1 if iso else max(len(sampled_steps), 1)
            have = [timing[k] for k in keys if k in timing]
            # launches, ms, work (FLOPs or bytes), bound ms - per step
            return tuple(sum(t[i] for t in have) / per for i in (0, 2, 3, 4))

        def mfma_entry(keys, label, traffic=None, peak_tfs=MFMA_BF16_PEAK_TFS):
            n, ms_ev, fl, bound_ms = event_sums(keys, False)
            _, ms_iso, _, _ = event_sums(keys, True)
            groups = sorted({EVENT_TO_GROUP[k] for k in keys if k in EVENT_TO_GROUP})
            ms_tr = sum(traced[g][0] for g in groups) if traced else 0.0
            n_tr = sum(traced[g][1] for g in groups) if traced else 0.0
            ms, mode = (ms_tr, "traced") if ms_tr > 0 else (ms_ev, "events")

            def tf(t_ms):
                return fl / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0

            ent = {"kernel": label, "bound": "mfma", "achieved": round(tf(ms), 1), "peak": peak_tfs, "unit": "TFLOP/s",
                   "frac": round(tf(ms) / peak_tfs, 4), "traffic": traffic, "mode": MODES[mode],
                   # per-launch roofline: sum over the launches of max(FLOPs / 2.5 PFLOP/s, compulsory bytes / 8 TB/s) over
                   # the measured time - most layers of this network are HBM-bound (a 128 -> 128 1x1 layer has 64 FLOP / B)
                   "bound_ms_per_step": round(bound_ms, 3), "bound_frac": round(bound_ms / ms, 4) if ms > 0 else 0.0,
                   "launches_per_step": round(n_tr if mode == "traced" else n, 1), "ms_per_step": round(ms, 3),
                   "algorithmic_tflop_per_step": round(fl / 1e12, 3),
                   "events": {"mode": MODES["events"], "ms_per_step": round(ms_ev, 3), "frac": round(tf(ms_ev) / peak_tfs, 4),
                              "launches_per_step": round(n, 1)},
                   "isolated": {"mode": MODES["isolated"], "ms_per_step": round(ms_iso, 3),
                                "frac": round(tf(ms_iso) / peak_tfs, 4),
                                "bound_frac": round(bound_ms / ms_iso, 4) if ms_iso > 0 else 0.0}}
            return ent

        def hbm_entry(key, label):
            if key not in timing or timing[key][2] <= 0:
                return None
            n, ms_ev, work, _ = event_sums((key,), False)
            per_launch = work / n
            g = EVENT_TO_GROUP[key]
            ms_tr, n_tr = traced[g] if traced else (0.0, 0.0)
            avg_ev = ms_ev / n
            avg, mode = (ms_tr / n_tr, "traced") if ms_tr > 0 and n_tr > 0 else (avg_ev, "events")
            gbs = per_launch / (avg * 1e-3) / 1e9
            return {"kernel": label, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(key, args.batch, lq, args.dtype),
                    "mode": MODES[mode], "algorithmic_bytes_per_launch": int(per_launch),
                    "launches_per_step": round(n_tr if mode == "traced" else n, 1), "avg_launch_ms": round(avg, 4),
                    "ms_per_step": round(avg * (n_tr if mode == "traced" else n), 3), "events_avg_launch_ms": round(avg_ev, 4)}

        fam_label = ("dense-conv implicit GEMMs of backbone + encoder: conv1x1_glds / conv3x3_ws / conv_igemm<3> (fwd + dgrad), "
                     "conv_wgrad1_glds / conv_wgrad3 + the deferred split reduction, stem_*")
        if args.dtype == "fp32":
            # config #2: every convolution / GEMM runs on the f32-input matrix cores - the family and its roof are those kernels'
            family = mfma_entry(F32_GROUPS, "fp32 convolutions and GEMMs on the f32-input matrix cores: conv_f32_kernel / wgrad_f32_kernel "
                                "(3x3, 2x2, strided), gemm_f32_big / gemm_f32_nt (1x1 convolutions and their weight gradients, linears, "
                                "attention products)", peak_tfs=MFMA_F32_PEAK_TFS)
        else:
            family = mfma_entry(MFMA_GROUPS, fam_label, traffic=conv_pmc_traffic())
            family["profile"] = profile_roofline()
        if traced:
            family["traced_kernel_ms_per_step"] = {g: round(v[0], 3) for g, v in traced.items()}
        kernels_ = [mfma_entry(("conv1x1",), "conv1x1_glds_kernel fwd+dgrad"),
                    mfma_entry(("conv3x3",), "conv3x3_ws_kernel / conv_igemm_kernel<3> fwd+dgrad"),
                    mfma_entry(("conv1x1_wgrad",), "conv_wgrad1_glds_kernel / conv_wgrad1_group_kernel"),
                    mfma_entry(("conv3x3_wgrad",), "3x3 weight gradient (conv_wgrad_kernel<3> / conv_wgrad3_*)"),
                    mfma_entry(("wgrad_reduce",), "multi_wgrad_reduce_kernel (split partial sums of all conv / linear weight gradients)"),
                    mfma_entry(("stem_conv", "stem_wgrad"), "stem_conv / stem_mfma / stem_dgrad_s2 / stem_wgrad"),
                    mfma_entry(("linear_wgrad",), "linear_wgrad_kernel (token-stream linears)"),
                    mfma_entry(("linear", "attention"), "linear_act / linear_ring / attention kernels (token streams)"),
                    mfma_entry(("miopen_conv",), "MIOpen convolutions (shapes the HIP weight-gradient kernel does not take)"),
                    hbm_entry("msda_fwd", "msda_fwd8_kernel (dfine_msda_fused_fwd)"),
                    hbm_entry("msda_bwd", "msda_bwd_pair_kernel (dfine_msda_fused_bwd_acc, packed-f16 accumulate)")]
        line = {
            "metric": "images/sec train step D-FINE-m 640x640 bs=32 at 1/2/4/8 MI355X",
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "median_ms_per_step": round(statistics.median(per_step), 3),
            "max_ms_per_step": round(max(per_step), 3), "slowest_step": per_step.index(max(per_step)),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"D-FINE-{args.model}{'+mask' if args.mask else ''} {args.img}x{args.img} bs={args.batch}/GPU full train step "
                                   "(fwd + Hungarian matcher/criterion + bwd + clip + AdamW + EMA), COCO-80 synthetic labels",
                       "global_batch": args.batch * world, "queries": lq, "parallelism": f"dp{world}",
                       "collective": f"RCCL all-reduce over {world} ranks" if world > 1 else "none",
                       "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None,
                       "ranks_seen": world, "per_rank_ms_per_step": [round(e / args.steps * 1e3, 3) for e in per_rank],
                       "hip_graph": bool(getattr(step, "hip_graph", False)), "instrumented_steps_in_timed_region": 0,
                       # eager steps run after the timed region: [0] un-instrumented (the first eager step of the process: grows
                       # the default-stream pool of the caching allocator), [1:] with HIP events around the reported launches
                       "eager_steps_after_timed_region": eager_ms},
            "roofline": family,
            "roofline_kernels": [k for k in kernels_ if k is not None and k.get("launches_per_step", 0) > 0],
        }
        if world == 1 and args.cpu_steps > 0:
            line["cpu_baseline"] = cpu_baseline(args.model, args.img, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
