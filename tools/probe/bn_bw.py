"""Effective bandwidth of the BatchNorm (+ReLU +LAB) forward / backward launches on the layer shapes of D-FINE-m at 640 / bs 32:
time per call and GB/s against the compulsory traffic (forward: read x, write y = 2 M; backward: read x, dy, write dx = 3 M) and
against what the two-pass kernels move (3 M / 5 M).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from custom_d_fine_amd import hip
dev = torch.device("cuda")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 32
shapes = [(24, 320), (12, 320), (24, 160), (32, 160), (48, 160), (96, 160), (96, 80), (64, 80), (192, 80), (384, 80), (384, 40), (128, 40),
          (768, 40), (768, 20), (256, 20), (1536, 20), (256, 80), (512, 80), (128, 80), (256, 40), (512, 40)]
tot_f = tot_b = tot_m = 0.0
print(f"{'C':>5} {'HW':>4} {'MB':>6} | {'fwd us':>7} {'GB/s(2M)':>8} | {'bwd us':>7} {'GB/s(3M)':>8}")
for C, H in shapes:
    x = torch.randn(B, C, H, H, device=dev).bfloat16()
    dy = torch.randn_like(x)
    g, b_ = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ls, lb = torch.tensor([1.2], device=dev), torch.tensor([0.1], device=dev)
    y, st = hip.bn_act_forward(x, g, b_, rm, rv, ls, lb, "relu", True, 0.1, 1e-5)
    tf = t(lambda: hip.bn_act_forward(x, g, b_, rm, rv, ls, lb, "relu", True, 0.1, 1e-5))
    tb = t(lambda: hip.bn_act_backward(x, dy, st, ls, "relu", True, True, True))
    M = x.numel() * 2 / 1e6
    print(f"{C:5d} {H:4d} {M:6.1f} | {tf:7.1f} {2 * M / tf * 1e3 / 1e3:8.0f} | {tb:7.1f} {3 * M / tb * 1e3 / 1e3:8.0f}")
