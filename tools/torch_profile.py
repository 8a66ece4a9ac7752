"""Steady-state per-kernel breakdown of the train step with torch.profiler (excludes MIOpen's
find phase, which pollutes a whole-process rocprofv3 trace).  GPU box only:
    python tools/torch_profile.py [--model m --batch 32 --img 640 --steps 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="m"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--img", type=int, default=640); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=4); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--rows", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step(a.model, a.img, dev, torch.bfloat16 if a.dtype == "bf16" else None)
images, targets = make_batch(a.batch, a.img, seed=42, device=dev)
for _ in range(a.warmup):
    step(images, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step(images, targets)
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter()-t0)/a.steps*1e3:.1f} ms/step")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(a.steps):
        step(images, targets)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 or getattr(e, "self_device_time_total", 0) > 0]
tot = sum(e.self_device_time_total for e in prof.key_averages())
print(f"total device time {tot/1e3/a.steps:.1f} ms/step")
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=a.rows, max_name_column_width=90))
