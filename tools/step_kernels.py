"""Top device kernels of one steady-state train step (torch.profiler), any bench configuration:
   SP_MODEL=x SP_IMG=960 SP_BATCH=8 SP_MASK=1 SP_DTYPE=bf16 python tools/step_kernels.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
MODEL, IMG, BATCH = os.environ.get("SP_MODEL", "m"), int(os.environ.get("SP_IMG", "640")), int(os.environ.get("SP_BATCH", "32"))
MASK, DT = os.environ.get("SP_MASK", "0") == "1", os.environ.get("SP_DTYPE", "bf16")
step = bench.build_step(MODEL, IMG, dev, torch.bfloat16 if DT == "bf16" else None, mask=MASK)
images, targets = make_batch(BATCH, IMG, seed=42, device=dev, with_masks=MASK)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA], record_shapes=False) as prof:
    for _ in range(2):
        step(images, targets)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda k: -k.device_time_total)
print(f"device time per step {sum(k.device_time_total for k in rows) / 2e3:.1f} ms")
for k in rows[:rows_n]:
    print(f"{k.device_time_total / 2e3:8.2f} ms  {k.count // 2:5d} x {k.device_time_total / max(k.count, 1):8.1f} us  {k.key[:120]}")
