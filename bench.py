"""Headline benchmark: images/sec of one full D-FINE-m 640x640 train step, bs=32 per GPU
(BASELINE.json metric / configs[2]) - fwd (bf16 autocast) + Hungarian matcher + criterion (fp32)
+ bwd + grad clip + AdamW + EMA on a device-resident synthetic batch.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      for the dominant hand-written kernel (the fused deformable-attention gather,
                dfine_msda_fused_fwd): algorithmic bytes per launch / mean launch time measured
                with HIP events on the launch stream inside the timed region
  cpu_baseline  the same train step through the CPU oracle backend ("port") on the host cores,
                D-FINE-m 640x640 at bs=2, bounded to a few steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LRS = {"n": (8e-4, 4e-4), "s": (2.5e-4, 6e-5), "m": (1.5e-4, 2e-5), "l": (1.6e-4, 1e-5), "x": (2e-4, 2e-6)}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_step(model_name, img, device, amp_dtype, num_classes=80, channels_last=False):
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep, wrap_data_parallel
    base_lr, backbone_lr = LRS[model_name]
    model = dfine.build_model(model_name, num_classes, False, str(device), img_size=[img, img]).train()
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    criterion = dfine.build_loss(model_name, num_classes, 0.0, False)
    ema = ModelEMA(model, 0.9998)
    fused = None
    opt = dfine.build_optimizer(model, lr=base_lr, backbone_lr=backbone_lr, betas=(0.9, 0.999),
                                weight_decay=1.25e-4, base_lr=base_lr)
    if device.type == "cuda" and os.environ.get("DFINE_FUSED_OPT", "1") == "1":
        # flat-buffer clip + AdamW + EMA kernels; data parallelism = one all-reduce of the flat grads
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
                     hip_graph=device.type == "cuda" and os.environ.get("DFINE_HIPGRAPH", "0") == "1")


def msda_algorithmic_bytes(batch, lq, heads=8, head_dim=32, points=12, elt=2):
    """SURVEY.md 8(d) per-image-per-layer figure x images of one launch (forward):
    gathered value reads Lq*H*P*4 corners*hd*elt + offsets Lq*H*P*2*elt + logits Lq*H*P*elt
    + reference boxes Lq*4*4 + output Lq*H*hd*elt."""
    per_img = (lq * heads * points * 4 * head_dim * elt + lq * heads * points * 2 * elt
               + lq * heads * points * elt + lq * 16 + lq * heads * head_dim * elt)
    return per_img * batch


def msda_pmc_traffic(batch, lq, dtype):
    """HBM bytes per launch of the gather kernel from the committed PMC passes (profiles/), or None
    when this run's shape differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_msda_pmc.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    sh = rec["shape"]
    if (sh["B"], sh["Lq"], sh["dtype"]) != (batch, lq, dtype):
        return None
    return rec["traffic_bytes_per_launch"]


def cpu_baseline(model_name, img, steps):
    """The same train step on the host through the oracle backend (kind 'port')."""
    from oracle import torch_backend
    from custom_d_fine_amd.dl.synthetic import make_batch
    torch_backend.install()
    try:
        bs = 2
        step = build_step(model_name, img, torch.device("cpu"), None)
        images, targets = make_batch(bs, img, seed=42)
        step(images, targets)  # warm-up (allocator, thread pools)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(images, targets)
        dt = time.perf_counter() - t0
    finally:
        torch_backend.uninstall()
    return {"value": round(bs * steps / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"D-FINE-{model_name} {img}x{img} bs={bs} fp32 full train step, 1 warm-up + {steps} timed steps "
                      f"({dt:.1f} s) through oracle/torch_backend.py on {os.cpu_count()} logical host CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="m")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU-baseline steps (0 = skip)")
    ap.add_argument("--channels-last", type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the HIP path and needs an MI355X (no CPU fallback)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from custom_d_fine_amd import hip
    from custom_d_fine_amd.dl.synthetic import make_batch
    if os.environ.get("DFINE_MIOPEN_BENCHMARK", "0") == "1":
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(42 + rank)
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    step = build_step(args.model, args.img, device, amp, channels_last=bool(args.channels_last))
    images, targets = make_batch(args.batch, args.img, seed=42 + rank, device=device)
    if args.channels_last:
        images = images.contiguous(memory_format=torch.channels_last)

    # W untimed warm-up steps; at least two untimed steps always run, because the first step measures the per-shape
    # conv plans (and MIOpen's find) and the second builds the batched weight-pack / bf16-shadow tables
    for _ in range(max(args.warmup, 2)):
        step(images, targets)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    hip.enable_timing(["dfine_msda_fused_fwd", "dfine_msda_fused_bwd"])
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(images, targets)
    fence()
    elapsed = time.perf_counter() - t0
    timing = hip.timing_summary()
    hip.disable_timing()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        max_t = max(len(t["labels"]) for t in targets)
        dn = 2 * max_t * max(100 // max_t, 1)
        lq = 300 + dn
        elt = 2 if args.dtype == "bf16" else 4
        n_fwd, ms_fwd = timing["dfine_msda_fused_fwd"]
        n_bwd, ms_bwd = timing["dfine_msda_fused_bwd"]
        algo = msda_algorithmic_bytes(args.batch, lq, elt=elt)
        achieved = algo / (ms_fwd * 1e-3) / 1e9 if ms_fwd > 0 else 0.0
        line = {
            "metric": "images/sec train step D-FINE-m 640x640 bs=32 at 1/2/4/8 MI355X",
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"D-FINE-{args.model} {args.img}x{args.img} bs={args.batch}/GPU full train step "
                                   "(fwd + Hungarian matcher/criterion + bwd + clip + AdamW + EMA), COCO-80 synthetic labels",
                       "global_batch": args.batch * world, "queries": lq, "parallelism": f"dp{world}"},
            "roofline": {"kernel": "msda_fwd8_kernel (dfine_msda_fused_fwd)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": msda_pmc_traffic(args.batch, lq, args.dtype),
                         "algorithmic_bytes_per_launch": algo, "launches": n_fwd,
                         "avg_launch_ms": round(ms_fwd, 4),
                         "bwd_avg_launch_ms": round(ms_bwd, 4), "bwd_launches": n_bwd},
        }
        if world == 1 and args.cpu_steps > 0:
            line["cpu_baseline"] = cpu_baseline(args.model, args.img, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
