"""Would two 3x3 weight gradients of the half-chip layers (128 -> 128 @40x40 / @20x20: 128 workgroups each) run side by side? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for C, H in ((128, 40), (128, 20), (128, 80), (256, 20)):
    xs = [torch.randn(32, C, H, H, device=dev).bfloat16() for _ in range(2)]
    dys = [torch.randn(32, C, H, H, device=dev).bfloat16() for _ in range(2)]
    hip.WGRAD_STREAM = False
    def one(i):
        return hip.conv_wgrad_bf16(xs[i], dys[i], 3)
    for _ in range(3): one(0); one(1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): one(0); one(1)
    b.record(); torch.cuda.synchronize()
    seq = a.elapsed_time(b) / 20 * 1e3
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1): one(0)
        with torch.cuda.stream(s2): one(1)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    b.record(); torch.cuda.synchronize()
    par = a.elapsed_time(b) / 20 * 1e3
    print(f"{C} -> {C} @{H}x{H}: two weight gradients one after the other {seq:6.1f} us, on two streams {par:6.1f} us")
