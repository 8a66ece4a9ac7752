import os, sys, torch
sys.path.insert(0, "/root/repo")
from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block
cuda = torch.device("cuda:0")
def run(link, eps_scale=1.0, light=True, B=12, H=80):
    os.environ["DFINE_BN_LINK"] = link
    kernels.reload_env()
    torch.manual_seed(3)
    blk = HG_Block(64, 32, 128, 3, residual=False, kernel_size=5 if light else 3, light_block=light, use_lab=True).to(cuda).train()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d): m.eps *= eps_scale
    x = torch.randn(B, 64, H, H, device=cuda).bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x)
    go = torch.randn(y.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(5)).to(y.dtype)
    y.backward(go)
    return [p.grad.float().clone() for p in blk.parameters()], [n for n, _ in blk.named_parameters()]
g0, names = run("0")
g1, _ = run("1")
g2, _ = run("0", 1.01)
for n, a, b, c in zip(names, g0, g1, g2):
    if a.numel() > 1 and a.dim() == 1:
        print(f"{n:40s} max|g| {a.abs().max():8.3f}  link diff {(a-b).abs().max():7.4f}   eps-perturbed diff {(a-c).abs().max():7.4f}")
