"""Per-shape cost of the dense convolutions (MIOpen, bf16) in one D-FINE-m bs=32 step (GPU box)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
os.environ["DFINE_MFMA_CONV"] = "0"   # let every dense conv go through nn.Conv2d so the hook sees it
os.environ["DFINE_ALLOW_LIBRARY"] = "1"   # (the product raises instead of dropping to MIOpen; this tool measures MIOpen)
from custom_d_fine_amd.d_fine import dfine

dev = torch.device("cuda", 0)
m = dfine.build_model("m", 80, False, "cuda", img_size=[640, 640]).train()
cfgs = collections.OrderedDict()
def hook(mod, inp, out):
    if mod.groups != 1: return
    key = (tuple(inp[0].shape), tuple(mod.weight.shape), mod.stride, mod.padding)
    cfgs[key] = cfgs.get(key, 0) + 1
hs = [c.register_forward_hook(hook) for c in m.modules() if isinstance(c, nn.Conv2d)]
from custom_d_fine_amd.dl.synthetic import make_batch
x, t = make_batch(32, 640, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    m(x, t)
for h in hs: h.remove()
rows = []
# CONV_SURVEY_BENCHMARK=1: let MIOpen search its solvers exhaustively (slow: seconds per layer);
# CONV_SURVEY_MAXCIN=n: only layers with <= n input channels (48 = the stem)
torch.backends.cudnn.benchmark = os.environ.get("CONV_SURVEY_BENCHMARK", "0") == "1"
max_cin = int(os.environ.get("CONV_SURVEY_MAXCIN", "100000"))
for (ishape, wshape, stride, pad), cnt in cfgs.items():
    if ishape[1] > max_cin:
        continue
    xi = torch.randn(ishape, device=dev, dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn(wshape, device=dev, dtype=torch.bfloat16, requires_grad=True)
    def fwd(): return F.conv2d(xi, w, None, stride, pad)
    y = fwd(); go = torch.randn_like(y)
    for _ in range(3): y = fwd(); y.backward(go)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): y = fwd()
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): y = fwd(); y.backward(go)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 10 - tf
    mine_f = mine_d = mine_w = float("nan")
    from custom_d_fine_amd import hip as H
    ks = wshape[2]
    if stride == (1, 1) and ks in (1, 3) and ishape[1] % 16 == 0 and wshape[0] % 16 == 0 and ishape[3] <= 160:
        w32 = w.detach().float()
        xi_c = xi.detach().contiguous()
        def mf(): return H.conv_forward_bf16(xi_c, H.conv_pack_weights(w32, False), wshape[0], ks)
        def md(): return H.conv_forward_bf16(go, H.conv_pack_weights(w32, True), wshape[1], ks)
        for _ in range(3): mf(); md()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): mf()
        torch.cuda.synchronize(); mine_f = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10): md()
        torch.cuda.synchronize(); mine_d = (time.perf_counter() - t0) / 10
        if H.conv_wgrad_supported(ishape[2], ishape[3], ks):
            for _ in range(3): H.conv_wgrad_bf16(xi_c, go, ks)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): H.conv_wgrad_bf16(xi_c, go, ks)
            torch.cuda.synchronize(); mine_w = (time.perf_counter() - t0) / 10
    flops = 2 * y.numel() * wshape[1] * wshape[2] * wshape[3]
    byts = 2 * (xi.numel() + y.numel())
    rows.append((cnt * (tf + tb), cnt, ishape, wshape, stride, tf * 1e3, tb * 1e3, flops / tf / 1e12, 2 * flops / max(tb, 1e-9) / 1e12, byts / 1e6, mine_f * 1e3, mine_d * 1e3, mine_w * 1e3))
rows.sort(reverse=True)
tot_f = sum(r[1] * r[5] for r in rows); tot_b = sum(r[1] * r[6] for r in rows)
print(f"dense convs: {sum(r[1] for r in rows)} calls, fwd {tot_f:.1f} ms, bwd {tot_b:.1f} ms")
for r in rows:
    print(f"{r[0]*1e3:7.2f} ms x{r[1]:2d} in{list(r[2])} w{list(r[3])} s{r[4][0]} fwd {r[5]:.3f} ms ({r[7]:.0f} TF) bwd {r[6]:.3f} ms ({r[8]:.0f} TF) io {r[9]:.0f} MB | HIP fwd {r[10]:.3f} dgrad {r[11]:.3f} wgrad {r[12]:.3f} ms")
import math
print("HIP fwd total (eligible): %.2f ms vs MIOpen fwd %.2f ms on the same layers; HIP dgrad total %.2f ms" % (
    sum(r[1]*r[10] for r in rows if not math.isnan(r[10])), sum(r[1]*r[5] for r in rows if not math.isnan(r[10])),
    sum(r[1]*r[11] for r in rows if not math.isnan(r[11]))))
