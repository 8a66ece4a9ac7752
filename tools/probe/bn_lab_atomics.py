"""Does the learnable-affine gradient (two same-address atomics per channel block) lengthen the BatchNorm backward? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B, C, H in [(32, 128, 40), (32, 512, 40), (32, 768, 40), (32, 1536, 20), (32, 256, 20), (32, 96, 160), (32, 384, 80)]:
    x = torch.randn(B, C, H, H, device=dev).bfloat16(); dy = torch.randn_like(x)
    g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ls, lb = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    _, st0 = hip.bn_act_forward(x, g, bt, torch.zeros(C, device=dev), torch.ones(C, device=dev), None, None, "relu", True, 0.1, 1e-5)
    _, st1 = hip.bn_act_forward(x, g, bt, torch.zeros(C, device=dev), torch.ones(C, device=dev), ls, lb, "relu", True, 0.1, 1e-5)
    t0 = t(lambda: hip.bn_act_backward(x, dy, st0, None, "relu", True, True, False))
    t1 = t(lambda: hip.bn_act_backward(x, dy, st1, ls, "relu", True, True, True))
    print(f"[{B},{C},{H},{H}]  backward without lab {t0:6.1f} us   with lab {t1:6.1f} us")
