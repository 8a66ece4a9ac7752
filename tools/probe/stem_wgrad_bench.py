"""Weight-gradient kernels of the StemBlock layers at D-FINE-m / 640 / bs 32: time per call and bytes / time (GPU box).
DFINE_STEM_WGRAD_KX=0: the column-per-lane kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from custom_d_fine_amd import hip
dev = torch.device("cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 32
# name, cin, cout, ks, stride, pad, H (input), Ho
for name, cin, cout, ks, st, pad, H, Ho in [("stem1", 3, 24, 3, 2, 1, 640, 320), ("stem2a", 24, 12, 2, 1, 0, 320, 320),
                                            ("stem2b", 12, 24, 2, 1, 0, 320, 320), ("stem3", 48, 24, 3, 2, 1, 320, 160)]:
    x = torch.randn(B, cin, H, H, device=dev).bfloat16()
    dy = torch.randn(B, cout, Ho, Ho, device=dev).bfloat16()
    us = t(lambda: hip.stem_wgrad(x, dy, ks, st, pad))
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    w = hip.stem_wgrad(x, dy, ks, st, pad)
    xr = x.float()
    if pad == 0 and ks == 2:
        xr = torch.nn.functional.pad(xr, (0, 1, 0, 1))
    wr = torch.nn.grad.conv2d_weight(xr[:4], (cout, cin, ks, ks), dy[:4].float(), stride=st, padding=pad)
    w4 = hip.stem_wgrad(x[:4].contiguous(), dy[:4].contiguous(), ks, st, pad)
    print(f"{name:7s} {us:7.1f} us  {mb:6.0f} MB  {mb / us / 1e3 * 1e3:5.2f} TB/s   rel err vs fp32 (4 images) {float((w4 - wr).abs().max() / wr.abs().max()):.2e}")
