"""Stand-alone durations of the fused residual / gate + LayerNorm kernels at the decoder's token-stream size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
rows, D = 15744, 256
a = torch.randn(rows, D, device=dev)
b16 = torch.randn(rows, D, device=dev).bfloat16()
gate = torch.randn(rows, 2 * D, device=dev).bfloat16()
w, bias = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
dy = torch.randn(rows, D, device=dev)
def run():
    for mode, bb, g in ((0, b16, None), (1, b16, None), (2, b16, gate)):
        y, mean, rstd, y16 = hip.ln_fused_forward(mode, a, bb, g, w, bias, 1e-5, 65504.0, with_bf16=True)
        hip.ln_fused_backward(mode, a, bb, g, w, mean, rstd, dy, 65504.0, True, True, g is not None, True)
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): run()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:4]:
    print(f"   {e.device_time_total / e.count:8.1f} us x{e.count // 5}  {e.key[:70]}")
