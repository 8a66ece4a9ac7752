"""Kernel profile of the inference forward (Torch_model's network, bf16, deployed) at one batch size.  GPU box only:
python tools/probe/infer_profile.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from custom_d_fine_amd.infer.torch_model import Torch_model
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tm = Torch_model("m", None, 80, 640, 640, half=True, hip_graph=False)
tm.model.deploy()
x = torch.rand(bs, 3, 640, 640, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        tm.model(x)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            tm.model(x)
        torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda k: -k.device_time_total)
tot = sum(k.device_time_total for k in rows) / 3e3
print(f"bs {bs}: device time per forward {tot:.2f} ms, {sum(k.count for k in rows) // 3} launches")
for k in rows[:32]:
    print(f"{k.device_time_total / 3e3:7.3f} ms {k.count // 3:4d} x {k.device_time_total / max(k.count, 1):7.1f} us  {k.key[:105]}")
print("by launch count:")
for k in sorted(prof.key_averages(), key=lambda k: -k.count)[:40]:
    print(f"{k.count // 3:4d} x {k.device_time_total / max(k.count, 1):7.1f} us  {k.key[:120]}")
