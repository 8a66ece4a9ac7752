"""Input gradient of every HG_Block of D-FINE-m at 640 x 640 on a LEAF input: gradient hand-offs on (default) vs off (DFINE_GRAD_FANIN=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine import dfine
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = dfine.build_model("m", 80, False, "cuda", img_size=[640, 640]).train()
bb = model.backbone
shapes = {}
hs = [m.register_forward_pre_hook((lambda n: lambda mod, a: shapes.__setitem__(n, tuple(a[0].shape)))(n)) for n, m in bb.named_modules() if type(m).__name__ == "HG_Block"]
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    bb(torch.rand(2, 3, 640, 640, device=dev))
for h in hs: h.remove()
for n, m in bb.named_modules():
    if type(m).__name__ != "HG_Block":
        continue
    res = {}
    for fanin in ("1", "0"):
        os.environ["DFINE_GRAD_FANIN"] = fanin
        kernels.reload_env()
        torch.manual_seed(1)
        x = torch.randn(shapes[n], device=dev).bfloat16().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        g = torch.randn(y.shape, device=dev, generator=torch.Generator(device="cuda").manual_seed(3)).to(y.dtype)
        y.backward(g)
        res[fanin] = None if x.grad is None else x.grad.float().clone()
        m.zero_grad(set_to_none=True)
    a, b = res["1"], res["0"]
    print(n, shapes[n], "fan-in grad:", "NONE" if a is None else f"cos {torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item():.5f} norm ratio {(a.norm() / b.norm()).item():.4f}")
