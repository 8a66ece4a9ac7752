"""Where does the eval-mode (frozen BatchNorm statistics) bf16 forward of backbone + encoder leave the fp32 one?  Relative error of
every ConvBNAct / module output, bf16 vs fp32, in module order (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = dfine.build_model("m", 80, False, "cuda", img_size=[640, 640]).train()
images, _ = make_batch(int(os.environ.get("B", "8")), 640, seed=42, device=dev)
body = torch.nn.Sequential(model.backbone, model.encoder)
bns = [m for m in body.modules() if isinstance(m, torch.nn.BatchNorm2d)]
for m in bns:
    m.momentum = 1.0
with torch.no_grad():
    body(images)
mode = os.environ.get("MODE", "eval")
if mode == "eval":
    body.eval()
outs = {}


def hook(name, store):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            store[name] = out.detach().float()
    return f


def run(amp):
    store = {}
    hs = [m.register_forward_hook(hook(n, store)) for n, m in body.named_modules() if type(m).__name__ in ("ConvBNAct", "LightConvBNAct", "ConvNormLayer", "VGGBlock", "HG_Block", "TransformerEncoderLayer", "StemBlock", "RepNCSPELAN4", "CSPLayer", "SCDown")]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        feats = body(images)
    for h in hs:
        h.remove()
    return store, [f.float() for f in feats]


sa, fa = run(False)
sb, fb = run(True)
for k in sa:
    a, b = sa[k], sb[k]
    print(f"{((a - b).norm() / a.norm()).item():8.4f}  |a| {a.abs().mean().item():9.4f}  {k}  {tuple(a.shape)}")
print("outputs:", [((a - b).norm() / a.norm()).item() for a, b in zip(fa, fb)])
