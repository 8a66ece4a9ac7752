"""Which ATen element-wise ops still run in a train step, by input shapes (torch.profiler, CPU+CUDA, record_shapes):
   python tools/step_aten_shapes.py [op substring, default add] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
pat = sys.argv[1] if len(sys.argv) > 1 else "add"
rows_n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
MODEL, IMG, BATCH = os.environ.get("SP_MODEL", "m"), int(os.environ.get("SP_IMG", "640")), int(os.environ.get("SP_BATCH", "32"))
MASK = os.environ.get("SP_MASK", "0") == "1"
step = bench.build_step(MODEL, IMG, dev, torch.bfloat16, mask=MASK)
images, targets = make_batch(BATCH, IMG, seed=42, device=dev, with_masks=MASK)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
A = torch.profiler.ProfilerActivity
with torch.profiler.profile(activities=[A.CPU, A.CUDA], record_shapes=True) as prof:
    step(images, targets)
    torch.cuda.synchronize()
rows = [k for k in prof.key_averages(group_by_input_shape=True) if pat in k.key and k.key.startswith("aten::")]
rows.sort(key=lambda k: -k.device_time_total)
for k in rows[:rows_n]:
    print(f"{k.device_time_total / 1e3:8.3f} ms {k.count:4d} x  {k.key:28s} {str(k.input_shapes)[:110]}")
