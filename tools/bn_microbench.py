"""Achieved bandwidth of the BN/act kernels (csrc/bnact.hip) per activation shape of D-FINE-m at the bench
batch: forward (stats 1R + apply 1R1W) and backward (reduce 2R + apply 2R1W).  GPU box only."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from custom_d_fine_amd import hip
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
m = dfine.build_model("m", 80, False, "cuda", img_size=[640, 640]).train()
shapes = collections.Counter()
hs = [b.register_forward_hook(lambda mod, i, o: shapes.update([tuple(o.shape)])) for b in m.modules() if isinstance(b, nn.BatchNorm2d)]
x, t = make_batch(32, 640, device=dev)
os.environ["DFINE_HIP_UNITS"] = "0"          # run the plain modules once so the hooks see every BN output shape
from custom_d_fine_amd import kernels
kernels.reload_env()
with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
    m.backbone(x) if False else m.encoder(m.backbone(x))
for h in hs: h.remove()
os.environ["DFINE_HIP_UNITS"] = "1"; kernels.reload_env()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us

tot_f = tot_b = ideal_f = ideal_b = 0.0
rows = []
for shape, cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * torch.Size(kv[0]).numel()):
    B, C, H, W = shape
    xx = torch.randn(shape, device=dev).to(torch.bfloat16)
    dy = torch.randn(shape, device=dev).to(torch.bfloat16)
    g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ls, lb = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    y, stats = hip.bn_act_forward(xx, g, bt, rm, rv, ls, lb, "relu", True, 0.1, 1e-5)
    tf = timeit(lambda: hip.bn_act_forward(xx, g, bt, rm, rv, ls, lb, "relu", True, 0.1, 1e-5))
    tb = timeit(lambda: hip.bn_act_backward(xx, dy, stats, ls, "relu", True, True, True))
    mb = xx.numel() * 2 / 1e6
    rows.append((cnt, shape, mb, tf, tb))
    tot_f += cnt * tf; tot_b += cnt * tb; ideal_f += cnt * 3 * mb / 4.5; ideal_b += cnt * 5 * mb / 4.5   # us at 4.5 TB/s
print(f"BN units: fwd {tot_f/1e3:.2f} ms (3 passes at 4.5 TB/s: {ideal_f/1e3:.2f}), bwd {tot_b/1e3:.2f} ms (5 passes: {ideal_b/1e3:.2f})")
for cnt, shape, mb, tf, tb in rows:
    print(f"x{cnt:2d} {list(shape)} {mb:6.1f} MB  fwd {tf:6.1f} us = {3*mb/tf:5.2f} TB/s   bwd {tb:6.1f} us = {5*mb/tb:5.2f} TB/s")
