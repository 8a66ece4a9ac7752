import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block, HG_Stage
dev = torch.device("cuda")
def run(flag, light, res):
    os.environ["DFINE_PARK_EAGER"] = flag; kernels.reload_env()
    torch.manual_seed(3)
    blk = HG_Block(128, 32, 128, 3, residual=res, kernel_size=5 if light else 3, light_block=light, use_lab=True, agg="se").to(dev).train()
    x = torch.randn(4, 128, 40, 40, device=dev).bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xx = x * 1.0
        y = blk(xx)
    go = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).to(y.dtype)
    y.backward(go.clone())
    return y.float(), x.grad.float(), [p.grad.float().clone() for p in blk.parameters()]
for light in (True, False):
    for res in (True, False):
        a = run("0", light, res); b = run("1", light, res)
        print(light, res, (a[0]-b[0]).abs().max().item(), (a[1]-b[1]).abs().max().item() / a[1].abs().max().item(),
              max(((p-q).abs().max()/ (p.abs().max()+1e-9)).item() for p, q in zip(a[2], b[2])))
