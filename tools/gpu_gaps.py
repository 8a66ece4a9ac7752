"""Where the GPU idles inside a steady-state train step: lists the largest gaps between consecutive
device kernels (torch.profiler timeline) with the kernels on either side, and the total idle time.
GPU box only:   python tools/gpu_gaps.py [--steps 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--rows", type=int, default=40); ap.add_argument("--min-us", type=float, default=15.0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(a.warmup):
    step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(a.steps):
        step(images, targets)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if str(e.device_type).endswith("CUDA")]
evs.sort(key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in evs)
span = evs[-1].time_range.end - evs[0].time_range.start
print(f"{len(evs)} device events over {a.steps} steps: span {span/1e3/a.steps:.2f} ms/step, busy {busy/1e3/a.steps:.2f} ms/step")
cpu = [e for e in prof.events() if str(e.device_type).endswith("CPU")]
cpu.sort(key=lambda e: e.time_range.start)
gaps = []
end = evs[0].time_range.end
prev = evs[0]
for e in evs[1:]:
    g = e.time_range.start - end
    if g > a.min_us:
        gaps.append((g, prev.name[:70], e.name[:70], (e.time_range.start - evs[0].time_range.start) / 1e3, end, e.time_range.start))
    if e.time_range.end > end:
        end = e.time_range.end
        prev = e
tot = sum(g[0] for g in gaps)
print(f"gaps > {a.min_us} us: {len(gaps)} totalling {tot/1e3/a.steps:.2f} ms/step")
small = span - busy - tot
print(f"remaining (sub-{a.min_us} us launch gaps): {small/1e3/a.steps:.2f} ms/step")
for g in sorted(gaps, reverse=True)[: a.rows]:
    print(f"{g[0]:9.1f} us at t={g[3]:8.2f} ms   after [{g[1]}]   before [{g[2]}]")

print("\nhost ops running inside the 8 largest gaps (top-level ops only):")
for g in sorted(gaps, reverse=True)[:8]:
    lo, hi = g[4], g[5]
    inside = [c for c in cpu if c.time_range.start >= lo - 50 and c.time_range.start < hi]
    top, last_end = [], -1
    for c in inside:                      # keep outermost ops
        if c.time_range.start >= last_end:
            top.append(c)
            last_end = c.time_range.end
    print(f"--- gap {g[0]:.0f} us at t={g[3]:.2f} ms: {len(inside)} host ops, outermost:")
    for c in top[:40]:
        print(f"      {c.time_range.start - lo:8.1f} us  +{c.time_range.end - c.time_range.start:7.1f}  {c.name[:80]}")
