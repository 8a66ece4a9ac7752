import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Stage
dev = torch.device("cuda")
class Two(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = HG_Stage(64, 32, 128, 1, 3, downsample=False, light_block=False, kernel_size=3, use_lab=True, agg="se")
        self.b = HG_Stage(128, 32, 256, int(os.environ.get("NB", "2")), 3, downsample=True, light_block=True, kernel_size=5, use_lab=True, agg="se")
    def forward(self, x):
        u = self.a(x)
        fan = kernels.GradFanIn() if kernels.grad_fanin_enabled(u) else None
        v = self.b(u, fanin=fan)
        return [kernels.park_grad(u, fan), v]
x0 = torch.randn(4, 64, 40, 40, device=dev).bfloat16()
def run(park):
    os.environ["DFINE_PARK_EAGER"] = park; kernels.reload_env()
    torch.manual_seed(3)
    m = Two().to(dev).train()
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = m(x * 1.0)
    gen = torch.Generator(device=dev).manual_seed(5)
    torch.autograd.backward(list(outs), [torch.randn(o.shape, device=dev, generator=gen).to(o.dtype) for o in outs])
    return x.grad.float(), {n: p.grad.float().clone() for n, p in m.named_parameters()}
cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
g0, p0 = run("0"); g1, p1 = run("1")
print("x", cos(g0, g1))
for n in p0:
    c = cos(p0[n], p1[n]) if p0[n].numel() > 1 else float((p0[n] - p1[n]).abs() / (p0[n].abs() + 1e-9))
    if p0[n].numel() > 1 and c < 0.999:
        print(n, tuple(p0[n].shape), round(c, 4))
