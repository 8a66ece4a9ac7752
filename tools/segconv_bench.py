"""Segmented (concat-in-place) 1x1 convolution vs torch.cat + the whole-tensor kernel, aggregation layers of D-FINE-m (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_d_fine_amd import hip as H
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for parts, cout, side in [((768, 128, 128, 128, 128), 384, 40), ((384, 128, 128, 128, 128), 256, 80), ((256, 256), 512, 40), ((128, 64, 64, 64, 32), 192, 80), ((512, 128, 128), 256, 40)]:
    xs = [torch.randn(32, c, side, side, device=dev).bfloat16() for c in parts]
    cin = sum(parts)
    w = torch.randn(cout, cin, 1, 1, device=dev)
    w2, w2d = H.conv_pack_weights(w, False), H.conv_pack_weights(w, True)
    y = torch.empty(32, cout, side, side, device=dev, dtype=torch.bfloat16)
    outs = [torch.empty_like(x) for x in xs]
    xc = torch.cat(xs, 1)
    print(f"{parts} -> {cout} @{side}: fwd seg {t(lambda: H.conv1x1_seg_forward(xs, w2, (y,))):7.1f} us | cat {t(lambda: torch.cat(xs, 1)):6.1f} + conv {t(lambda: H.conv_forward_bf16(xc, w2, cout, 1)):7.1f}"
          f" || dgrad seg {t(lambda: H.conv1x1_seg_forward((y,), w2d, outs)):7.1f} | whole {t(lambda: H.conv_forward_bf16(y, w2d, cin, 1)):7.1f}"
          f" || wgrad seg {t(lambda: H.conv1x1_seg_wgrad(xs, y)):7.1f} | whole {t(lambda: H.conv_wgrad_bf16(xc, y, 1)):7.1f}")
