import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn, torch.nn.functional as F
from custom_d_fine_amd import kernels
dev = torch.device("cuda", 0)
for (cin, cout, k, s, p, H, W) in [(48, 32, 3, 2, 1, 160, 160), (32, 48, 1, 1, 0, 80, 80), (24, 32, 3, 2, 1, 160, 160), (48, 32, 3, 2, 1, 80, 80), (48, 32, 3, 1, 1, 160, 160)]:
    torch.manual_seed(0)
    conv = nn.Conv2d(cin, cout, k, s, p, bias=False).to(dev)
    x = torch.randn(2, cin, H, W, device=dev, requires_grad=True)
    y = kernels.conv_f32(x, conv)
    go = torch.randn_like(y)
    y.backward(go)
    xr = x.detach().double().requires_grad_(True); wr = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p); yr.backward(go.double())
    cos = lambda a, b: F.cosine_similarity(a.double().flatten(), b.flatten(), dim=0).item()
    print((cin, cout, k, s, H), "y", (y.double() - yr).abs().max().item() / yr.abs().max().item(), "dx", cos(x.grad, xr.grad), "dw", cos(conv.weight.grad, wr.grad),
          (conv.weight.grad.double() - wr.grad).abs().max().item() / wr.grad.abs().max().item())
