"""1x1 weight-gradient kernels (+ the split reduction) at the layer shapes of D-FINE-m bs=32.  DFINE_WGRAD1_GLDS=0 selects the first-generation register-staged kernel for an A/B run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_d_fine_amd import hip as H

dev = torch.device("cuda", 0)
SHAPES = [(128, 128, 40), (128, 128, 80), (128, 128, 20), (384, 768, 40), (1280, 384, 40), (768, 256, 40), (512, 512, 80), (768, 256, 80),
          (160, 48, 160), (256, 128, 80), (512, 512, 40), (256, 128, 40), (256, 256, 20), (352, 192, 80), (384, 256, 80), (48, 96, 160),
          (768, 128, 40), (192, 384, 80), (896, 384, 40), (256, 256, 80), (256, 256, 40), (1792, 768, 20), (768, 256, 20), (768, 1536, 20),
          (1536, 256, 20), (512, 512, 20)]
tot = 0.0
for cin, cout, side in SHAPES:
    x = torch.randn(32, cin, side, side, device=dev).bfloat16()
    dy = torch.randn(32, cout, side, side, device=dev).bfloat16()
    for _ in range(3):
        H.conv_wgrad_bf16(x, dy, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        H.conv_wgrad_bf16(x, dy, 1)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    tot += us
    fl = 2.0 * 32 * side * side * cin * cout
    io = 2.0 * 32 * side * side * (cin + cout)
    print(f"{cin:5d} -> {cout:5d} @{side:3d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {io/us/1e3:7.1f} GB/s")
print(f"sum {tot:.0f} us  (DFINE_WGRAD1_GLDS={os.environ.get('DFINE_WGRAD1_GLDS', '1')})")
