import torch, torch.nn.functional as F, sys
sys.path.insert(0,'/root/repo')
from custom_d_fine_amd import kernels, hip
torch.manual_seed(0)
cuda=torch.device('cuda')
lin = torch.nn.Linear(256, 132).to(cuda)
x = torch.randn(16, 300, 256, device=cuda, requires_grad=True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = kernels.linear(x, lin.weight, lin.bias)
    y2 = F.relu(y, inplace=True)
go = torch.randn_like(y2)
y2.backward(go)
g = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
x.grad=None; lin.zero_grad()
with torch.autocast("cuda", dtype=torch.bfloat16):
    F.relu(F.linear(x, lin.weight, lin.bias)).backward(go)
for n,a,b in zip("xwb",g,(x.grad, lin.weight.grad, lin.bias.grad)):
    d=(a-b).abs(); i=d.argmax(); print(n, d.max().item(), b.abs().max().item(), a.flatten()[i].item(), b.flatten()[i].item(), i.item(), a.shape)
# direct dgrad check
d2 = (go*(y2>0)).bfloat16().reshape(-1,132)
wt = lin.weight.detach().t().bfloat16().contiguous()
dx = hip.linear_act(d2, wt, None, 0, out_f32=True)
ref = d2.float() @ lin.weight.detach().bfloat16().float()
dd=(dx-ref).abs(); print("direct", dd.max().item(), (dd>0.03).sum().item(), torch.nonzero(dd>0.03)[:5].tolist())
