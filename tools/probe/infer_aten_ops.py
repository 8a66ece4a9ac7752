import os, sys
sys.path.insert(0, "/root/repo")
import torch
from custom_d_fine_amd.infer.torch_model import Torch_model
tm = Torch_model("m", None, 80, 640, 640, half=True, hip_graph=False)
tm.model.deploy()
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 3, 640, 640, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        tm.model(x)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        tm.model(x)
        torch.cuda.synchronize()
rows = [k for k in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6) if k.key.startswith("aten::") and k.self_device_time_total > 0]
rows.sort(key=lambda k: -k.self_device_time_total)
for k in rows[:14]:
    print(f"{k.self_device_time_total:8.1f} us {k.count:3d} x {k.key:22s} {str(k.input_shapes)[:90]}")
    for fr in (k.stack or [])[:6]:
        if "custom_d_fine_amd" in fr:
            print("        ", fr[-110:])
