"""Which ATen operators (by input shapes) launch the element-wise / reduction kernels of a train step: config #2 (fp32 D-FINE-s)
by default, any other through the arguments.
GPU box only:   python tools/probe/aten_ops.py [model img batch dtype(fp32|bf16) mask(0|1)]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
a = sys.argv[1:] + ["s", "640", "16", "fp32", "0"][len(sys.argv) - 1:]
MODEL, IMG, BATCH, MASK = a[0], int(a[1]), int(a[2]), a[4] == "1"
step = bench.build_step(MODEL, IMG, dev, None if a[3] == "fp32" else torch.bfloat16, mask=MASK)
step.hip_graph = False
images, targets = make_batch(BATCH, IMG, seed=42, device=dev, with_masks=MASK)
for _ in range(3):
    step(images, list(targets))
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(images, list(targets))
    torch.cuda.synchronize()
rows = [k for k in prof.key_averages(group_by_input_shape=True) if k.key.startswith("aten::") and k.self_device_time_total > 0]
rows.sort(key=lambda k: -k.self_device_time_total)
tot = sum(k.self_device_time_total for k in rows)
print(f"ATen device time {tot / 1e3:.2f} ms per step")
for k in rows[:45]:
    print(f"{k.self_device_time_total / 1e3:7.3f} ms {k.count:4d} x  {k.key:28s} {str(k.input_shapes)[:120]}")
