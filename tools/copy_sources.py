"""Where the small device copies / fills of a steady-state train step come from: aten::copy_ / fill_ / zero_ calls grouped
by Python call site (GPU box)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True, record_shapes=True) as prof:
    step(images, targets)
    torch.cuda.synchronize()
want = {"aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::add", "aten::add_"}
only = set(os.environ.get("OPS", "aten::copy_,aten::fill_").split(","))
rows = collections.Counter()
dtime = collections.Counter()
for e in prof.events():
    if e.name in only and e.device_time_total > 0:
        frames = [f for f in (e.stack or []) if "custom_d_fine_amd" in f or "bench.py" in f][:2]
        key = (e.name, str(e.input_shapes)[:50], " <- ".join(f.split("custom_d_fine_amd/")[-1][:70] for f in frames))
        rows[key] += 1
        dtime[key] += e.device_time_total
for key, t in sorted(dtime.items(), key=lambda kv: -kv[1])[:int(os.environ.get("TOP", "60"))]:
    print(f"{t / 1e3:7.3f} ms x{rows[key]:3d} {key[0]:12s} {key[1]:50s} {key[2]}")
print("total", sum(dtime.values()) / 1e3, "ms,", sum(rows.values()), "calls")
