"""Stand-alone cost of the depthwise-convolution and stem weight-gradient launches at the D-FINE-m bs=32 shapes (kernel durations
from the profiler) against the bytes they have to move - in the train step they run on the side stream next to the main chain, where
every kernel is slower than alone.   GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from custom_d_fine_amd import hip

dev = torch.device("cuda", 0)


def timed(fn, pat):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_time > 0]
    out = {}
    for e in evs:
        out[e.name[:60]] = out.get(e.name[:60], 0.0) + e.device_time / 5
    return out


print("depthwise layers (x bf16 [32, C, H, W]): kernel -> us stand-alone; bytes = x + dy (+ dx)")
for C, H, K, s in [(128, 40, 5, 1), (256, 20, 5, 1), (96, 160, 3, 2), (384, 80, 3, 2), (768, 40, 3, 2), (256, 80, 3, 2), (256, 40, 3, 2)]:
    x = torch.randn(32, C, H, H, device=dev).bfloat16()
    w = torch.randn(C, 1, K, K, device=dev)
    y = hip.dwconv_forward(x, w, s, K // 2)
    dy = torch.randn_like(y)
    mb = (x.numel() + y.numel()) * 2 / 1e6
    for name, fn in (("fwd", lambda: hip.dwconv_forward(x, w, s, K // 2)),
                     ("dgrad", lambda: hip.dwconv_backward(x, w, dy, s, K // 2, need_dx=True, need_dw=False)),
                     ("wgrad", lambda: hip.dwconv_backward(x, w, dy, s, K // 2, need_dx=False, need_dw=True))):
        t = timed(fn, "dwconv")
        desc = ", ".join(f"{k.split('(')[0].replace('void dfine::', '')[:34]} {v:.1f}" for k, v in t.items())
        tot = sum(t.values())
        print(f"  C {C:4d} {H:3d}x{H:<3d} k{K} s{s} {name:6s} {tot:7.1f} us  ({mb:6.1f} MB -> {mb / tot * 1e3 / 1e3:5.2f} TB/s)  {desc}")
