"""Which ATen operators (and input shapes) still run in a steady-state train step, by device time (GPU box)."""
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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(images, targets)
    torch.cuda.synchronize()
rows = []
for k in prof.key_averages(group_by_input_shape=True):
    if k.key.startswith("aten::") and k.self_device_time_total > 0:
        rows.append((k.self_device_time_total / 1e3, k.count, k.key, str(k.input_shapes)[:110]))
rows.sort(reverse=True)
print(f"aten ops with device time: {sum(r[0] for r in rows):.2f} ms")
for r in rows[:int(os.environ.get('TOP', '45'))]:
    print(f"{r[0]:6.3f} ms x{r[1]:3d} {r[2]:28s} {r[3]}")
