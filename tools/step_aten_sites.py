"""Which source lines of this package issue the remaining ATen ops of a train step (torch.profiler with_stack):
   python tools/step_aten_sites.py cat zeros stack add_ to copy_"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
pats = sys.argv[1:] or ["cat", "zeros", "stack"]
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
A = torch.profiler.ProfilerActivity
with torch.profiler.profile(activities=[A.CPU, A.CUDA], with_stack=True) as prof:
    step(images, targets)
    torch.cuda.synchronize()
sites = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.name[6:] not in pats:
        continue
    frame = next((f for f in (e.stack or []) if "custom_d_fine_amd" in f or "bench.py" in f), "(autograd engine / no python frame)")
    frame = frame.split("custom_d_fine_amd/")[-1]
    k = (e.name, frame[:110])
    sites[k][0] += 1
    sites[k][1] += e.device_time_total if hasattr(e, "device_time_total") else 0.0
for (name, frame), (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{n:4d} x {t / 1e3:7.3f} ms  {name:14s} {frame}")
