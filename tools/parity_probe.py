"""Ad-hoc parity probe (container only): reference vs this build on CPU, same weights/inputs."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from ref_import import import_reference
ref = import_reference()
from oracle import torch_backend; torch_backend.install()
from custom_d_fine_amd.d_fine import dfine as mine

size = sys.argv[1] if len(sys.argv) > 1 else "n"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 320
torch.manual_seed(0)
rm = ref.dfine.build_model(size, 80, False, "cpu", img_size=[img, img])
mm = mine.build_model(size, 80, False, "cpu", img_size=[img, img])
sd = rm.state_dict()
# randomise every tensor so zero-initialised heads do not hide errors
g = torch.Generator().manual_seed(1)
for k, v in sd.items():
    if v.dtype.is_floating_point and "anchors" not in k and "running_var" not in k and k not in ("decoder.up","decoder.reg_scale") and "num_points_scale" not in k:
        if v.abs().sum() == 0:
            v.copy_(torch.randn(v.shape, generator=g) * 0.02)
    if "running_var" in k: v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    if "running_mean" in k: v.copy_(torch.randn(v.shape, generator=g) * 0.1)
rm.load_state_dict(sd)
missing = mm.load_state_dict(sd, strict=True)
print("state_dict keys equal:", list(mm.state_dict().keys()) == list(sd.keys()), len(sd))
x = torch.rand(2, 3, img, img, generator=g)
rm.eval(); mm.eval()
with torch.no_grad():
    a = rm(x); b = mm(x)
for k in a: print("eval", k, (a[k]-b[k]).abs().max().item())

targets = [{"labels": torch.tensor([1, 5, 7]), "boxes": torch.tensor([[.3,.4,.2,.2],[.6,.5,.3,.25],[.5,.5,.1,.3]])},
           {"labels": torch.tensor([2, 9]), "boxes": torch.tensor([[.4,.4,.3,.2],[.7,.6,.2,.25]])}]
rm.train(); mm.train()
rc = ref.dfine.build_loss(size, 80, 0.0, False); mc = mine.build_loss(size, 80, 0.0, False)
torch.manual_seed(5); oa = rm(x, targets); 
torch.manual_seed(5); ob = mm(x, targets)
def cmp(a, b, pre=""):
    for k in a:
        if isinstance(a[k], torch.Tensor) and a[k].dtype.is_floating_point:
            print("train", pre+k, (a[k]-b[k]).abs().max().item())
        elif isinstance(a[k], list):
            for i,(u,v) in enumerate(zip(a[k], b[k])): cmp(u, v, f"{pre}{k}[{i}].")
        elif isinstance(a[k], dict) and k != "dn_meta" and k != "enc_meta": cmp(a[k], b[k], pre+k+".")
cmp(oa, ob)
la = rc(oa, targets); lb = mc(ob, targets)
print(len(la), len(lb), set(la)==set(lb))
worst = 0
for k in la:
    d = abs(la[k].item()-lb[k].item()); worst = max(worst, d)
    if d > 1e-4: print("LOSS DIFF", k, la[k].item(), lb[k].item())
print("worst loss diff", worst, "total", sum(v.item() for v in la.values()), sum(v.item() for v in lb.values()))
sum(la.values()).backward(); sum(lb.values()).backward()
gw = 0
for (n1,p1),(n2,p2) in zip(rm.named_parameters(), mm.named_parameters()):
    if p1.grad is None: assert p2.grad is None, n1; continue
    d = (p1.grad-p2.grad).abs().max().item(); s = p1.grad.abs().max().item()
    if d > 1e-3*max(s,1e-3): print("GRAD DIFF", n1, d, s)
    gw = max(gw, d)
print("worst grad diff", gw)

if "--debug-eval" in sys.argv:
    rm.eval(); mm.eval()
    with torch.no_grad():
        fa = rm.backbone(x); fb = mm.backbone(x)
        print("bb", [(u-v).abs().max().item() for u,v in zip(fa,fb)])
        ea = rm.encoder(fa); eb = mm.encoder(fa)
        print("enc", [(u-v).abs().max().item() for u,v in zip(ea,eb)])
        da = rm.decoder(ea); db = mm.decoder(ea)
        print("dec", [(da[k]-db[k]).abs().max().item() for k in da])
        d = (da["pred_logits"]-db["pred_logits"]).abs().amax(-1)
        print("queries differing:", (d>1e-4).sum().item(), "of", d.numel())
    with torch.no_grad():
        a = rm(x); b = mm(x)
    for k in a: print("eval-after-train", k, (a[k]-b[k]).abs().max().item())
    import custom_d_fine_amd.kernels as K
    sel = {}
    orig_topk = torch.topk
    def spy(score, k, dim=-1):
        r = orig_topk(score, k, dim=dim); sel.setdefault("s", []).append((score.clone(), r.indices.clone())); return r
    torch.topk = spy
    with torch.no_grad():
        a = rm(x); b = mm(x)
    torch.topk = orig_topk
    (sa, ia), (sb, ib) = sel["s"][0], sel["s"][1]
    print("score diff", (sa-sb).abs().max().item(), "idx equal", torch.equal(ia, ib))
    for bi in range(2):
        print("set equal", set(ia[bi].tolist()) == set(ib[bi].tolist()), "margin at 300:", (sa[bi].sort(descending=True).values[299]-sa[bi].sort(descending=True).values[300]).item())
        diff = (ia[bi]!=ib[bi]).nonzero().flatten()
        print(" first diffs at ranks", diff[:6].tolist(), [ (sa[bi][ia[bi][r]].item(), sa[bi][ib[bi][r]].item()) for r in diff[:3].tolist()])
