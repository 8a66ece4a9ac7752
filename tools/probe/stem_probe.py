import torch, sys, os
sys.path.insert(0, "/root/repo")
from custom_d_fine_amd import hip
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
x = torch.randn(32, 24, 320, 320, device=dev).bfloat16()
w = torch.randn(12, 24, 2, 2, device=dev)
wp = hip.stem_pack_weights(w, 0)
for _ in range(3): hip.stem_conv(x, wp, 12, 2, 1, 0, (320, 320))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): hip.stem_conv(x, wp, 12, 2, 1, 0, (320, 320))
    torch.cuda.synchronize()
for e in prof.key_averages():
    if e.device_time_total > 0: print(os.environ.get("DFINE_STEM_DBG", "0"), e.key[:60], e.device_time_total / e.count)
