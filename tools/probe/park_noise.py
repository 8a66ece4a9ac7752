"""How far do the backbone + encoder gradients move when only the ORDER of the bf16 gradient sums changes?  Eager runs of
D-FINE-n / m at 320 with (a) the default hand-offs, (b) DFINE_PARK_EAGER=1 (the captured segments' extra hand-offs: three-term sums
in a different association), (c) DFINE_GRAD_FANIN=0 (autograd adds everything).  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from custom_d_fine_amd import kernels
from custom_d_fine_amd.dl.engine import _BackboneEncoder
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "n"
torch.manual_seed(0)
step = bench.build_step(name, 320, dev, torch.bfloat16)
images, _ = make_batch(4, 320, seed=3, device=dev)
be = _BackboneEncoder(step.model.backbone, step.model.encoder)
params = dict(be.named_parameters())
def run(env):
    for k in ("DFINE_PARK_EAGER", "DFINE_GRAD_FANIN"):
        os.environ.pop(k, None)
    os.environ.update(env); kernels.reload_env()
    for p in params.values():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = be(images)
    g = torch.Generator(device="cpu").manual_seed(5)
    gouts = [(torch.randn(f.shape, generator=g) * 1e-2).to(dev, torch.bfloat16) for f in feats]
    torch.autograd.backward(feats, gouts)
    step.fused._uses.clear() if hasattr(step, "fused") else None
    return {n: p.grad.float().clone() for n, p in params.items() if p.grad is not None}
step.fused = None if not hasattr(step, "fused") else step.fused
os.environ["DFINE_FUSED_DEFER"] = "0"
a = run({}); b = run({"DFINE_PARK_EAGER": "1"}); c = run({"DFINE_GRAD_FANIN": "0"}); a2 = run({})
def worst(x, y, tag):
    rows = sorted(((float((x[n] - y[n]).abs().max() / (x[n].abs().max() + 1e-12)), n) for n in x if n in y), reverse=True)[:4]
    print(tag, [(round(r, 4), n[-40:]) for r, n in rows])
worst(a, a2, "default vs default   ")
worst(a, b, "default vs park-eager")
worst(a, c, "default vs no fan-in ")
worst(b, c, "park-eager vs no fan-in")
