import os, sys, collections
sys.path.insert(0, "/root/repo")
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("s", 640, dev, None)
images, targets = make_batch(16, 640, seed=42, device=dev)
for _ in range(3):
    step(images, targets)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        step(images, targets)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda k: -k.device_time_total)
tot = sum(k.device_time_total for k in rows) / 2e3
print(f"device time per step {tot:.1f} ms")
for k in rows[:28]:
    print(f"{k.device_time_total/2e3:8.2f} ms  {k.count//2:5d} x {k.device_time_total/max(k.count,1):8.1f} us  {k.key[:110]}")
