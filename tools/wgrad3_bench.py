"""Stand-alone timing of the 3x3 weight-gradient kernel on the D-FINE-m layer shapes (GPU box): kernel time from torch.profiler.
   python tools/wgrad3_bench.py            DFINE_W3_ABLATE=1 (no loads after the first unit) / 2 (no MFMAs) for experiments"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from custom_d_fine_amd import hip

SHAPES = [((32, 128, 20, 24), 128, 6), ((32, 128, 40, 40), 128, 12), ((32, 128, 80, 80), 128, 6), ((32, 64, 80, 80), 64, 3),
          ((32, 96, 80, 80), 64, 1), ((32, 32, 160, 160), 32, 4)]
dev = torch.device("cuda", 0)
tot = 0.0
for xs, cout, n in SHAPES:
    x = torch.randn(xs, device=dev).bfloat16()
    dy = torch.randn(xs[0], cout, xs[2], xs[3], device=dev).bfloat16()
    for _ in range(3):
        hip.conv_wgrad_bf16(x, dy, 3, partials=True)
    hip.side_join()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            ws, meta = hip.conv_wgrad_bf16(x, dy, 3, partials=True)
        hip.side_join()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if "conv_wgrad" in e.name]
    us = sum(e.device_time for e in evs) / max(1, len(evs))
    fl = 2.0 * xs[0] * xs[2] * xs[3] * xs[1] * cout * 9
    tot += us * n
    print(f"{str(xs):>22} -> {cout:4d} x{n:2d} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s  splits {meta[0]:4d}  {evs[0].name[:60] if evs else ''}")
print(f"sum over the step: {tot / 1e3:.3f} ms   (DFINE_W3_ABLATE={os.environ.get('DFINE_W3_ABLATE', '0')}, DFINE_WGRAD3_ROWS={os.environ.get('DFINE_WGRAD3_ROWS', '1')})")
