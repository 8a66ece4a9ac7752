"""Backbone + encoder forward + backward of D-FINE-m (bs 32, 640 x 640, bf16) in isolation: eager launch sequence against the
captured graphs of dl/engine.GraphedSegment, with and without the side stream inside the capture (GPU box).
    python tools/graph_probe8.py            # DFINE_GRAPH_SIDE from the environment
Prints host time of the enqueue and total (synchronised) time of forward, backward and both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd import hip, kernels
from custom_d_fine_amd.dl.engine import GraphedSegment, _BackboneEncoder
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
fused = step.fused
images, targets = make_batch(32, 640, seed=42, device=dev)
be = _BackboneEncoder(step.model.backbone, step.model.encoder)
kernels.defer_bn_counters(True)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    shapes = [f.shape for f in be(images)]
gouts = [(torch.randn(s, device=dev) * 1e-2).to(torch.bfloat16) for s in shapes]


def sync():
    torch.cuda.synchronize()


def measure(fwd, bwd, label, reps=6):
    for _ in range(2):
        f = fwd(); bwd(f)
    rows = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter()
        f = fwd(); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        bwd(f); t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t0, t3 - t2, t4 - t2))
        sync(); t0 = time.perf_counter(); f = fwd(); bwd(f); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        rows[-1] += (t1 - t0, t2 - t0)
    best = [min(r[i] for r in rows) * 1e3 for i in range(6)]
    print(f"{label:34s} fwd host {best[0]:6.2f} total {best[1]:6.2f} | bwd host {best[2]:6.2f} total {best[3]:6.2f} | "
          f"fwd+bwd unsynchronised: host {best[4]:6.2f} total {best[5]:6.2f} ms", flush=True)


def eager_fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return be(images)


def eager_bwd(feats):
    torch.autograd.backward(feats, gouts)
    fused._collect_grads()
    fused._uses.clear()
    fused.flat_grad.zero_()


modes = os.environ.get("PROBE8_MODES", "e1,e0,g,g1,g0").split(",")
for ws in (True, False):
    if f"e{int(ws)}" not in modes:
        continue
    hip.WGRAD_STREAM = ws
    measure(eager_fwd, eager_bwd, f"eager (side stream {int(ws)})")
hip.WGRAD_STREAM = True

for side in ("env", "1", "0"):
    if f"g{side}" not in modes and not (side == "env" and "g" in modes):
        continue
    if side != "env":
        os.environ["DFINE_GRAPH_SIDE"] = "fork" if side == "1" else "0"
    seg = GraphedSegment(be, (images,), amp_dtype=torch.bfloat16, fused=fused)

    def g_fwd():
        return seg(images)

    def g_bwd(feats):
        torch.autograd.backward(feats, gouts)
        fused.flat_grad.zero_()
    measure(g_fwd, g_bwd, f"graph (side stream {os.environ.get('DFINE_GRAPH_SIDE', 'dual')} {os.environ.get('DFINE_GRAPH_CHUNK', '')})")
    del seg
