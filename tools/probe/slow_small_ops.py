"""ATen ops of one train step whose device time is > 30 us, with input shapes / dtypes (torch.profiler, record_shapes) - GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(5):
    step(images, list(targets))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(images, list(targets))
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if e.key.startswith("aten::") and e.count and t / e.count > 25:
        rows.append((t / e.count, e.count, e.key, e.input_shapes))
for t, c, n, s in sorted(rows, reverse=True)[:30]:
    print(f"{t:8.1f} us x {c:3d}  {n:28s} {s}")
