"""stem3 (48 -> 24, 3x3 / stride 2 over two 24-channel maps, 320 x 320, batch 32) forward and data gradient: the row-streaming
MFMA kernels of csrc/stem3.hip against the direct kernels (DFINE_STEM3_ROWS=0), per-kernel device time (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
B, ch, cout, H, W = 32, 24, 24, 320, 320
xa = torch.randn(B, ch, H, W, device=dev).bfloat16()
xb = torch.randn(B, ch, H, W, device=dev).bfloat16()
w = torch.randn(cout, 2 * ch, 3, 3, device=dev) / (2 * ch * 9) ** 0.5
wp, wq = hip.stem_pack_weights(w, 0), hip.stem_pack_weights(w, 2)
go = torch.randn(B, cout, H // 2, W // 2, device=dev).bfloat16()
f = lambda: (hip.stem_conv2(xa, xb, wp, cout, 3, 2, 1, (H // 2, W // 2)), hip.stem_dgrad_s2_2(go, wq, ch, ch))
for _ in range(3): f()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(10): f()
    torch.cuda.synchronize()
mb = 2.0 * B * (2 * ch * H * W + cout * H * W / 4) / 1e6
for k in prof.key_averages():
    if "stem" in k.key:
        t = k.device_time_total / k.count
        print(f"DFINE_STEM3_ROWS={os.environ.get('DFINE_STEM3_ROWS', '1')}  {k.key[:60]:60s} {t:7.1f} us  {mb / t * 1e-6 * 1e6 / 1e3:6.2f} TB/s")
