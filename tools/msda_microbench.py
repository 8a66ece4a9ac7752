"""Stand-alone launches of the deformable-attention kernels at the bench shapes (D-FINE-m, 640x640,
bs=32, Lq=492, bf16) - used under `rocprofv3 --pmc ...` to read HBM traffic per launch (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_d_fine_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, Lq, H, D, L = 32, 492, 8, 32, 8400
shapes, points = ((80, 80), (40, 40), (20, 20)), (3, 6, 3)
value = torch.randn(B, L, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
ref = torch.cat([torch.rand(B, Lq, 2, device=dev), torch.rand(B, Lq, 2, device=dev) * 0.3 + 0.02], -1)
off = (torch.randn(B, Lq, H, 12, 2, device=dev) * 2).bfloat16().requires_grad_(True)
lg = torch.randn(B, Lq, H, 12, device=dev).bfloat16().requires_grad_(True)
go = torch.randn(B, Lq, H * D, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    kernels.msda_fused(value, shapes, ref, off, lg, points, 0.5).backward(go)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(n):
    ev[0].record(); o = kernels.msda_fused(value, shapes, ref, off, lg, points, 0.5); ev[1].record()
    o.backward(go); ev[2].record(); torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
import bench
algo = bench.msda_algorithmic_bytes(B, Lq, elt=2)
print(f"msda fused fwd {tf/n*1e3:.1f} us ({algo/(tf/n*1e-3)/1e9:.0f} GB/s algorithmic), bwd (incl. zero-fill + cast) {tb/n*1e3:.1f} us")
