"""Every fp32 GEMM / 3x3-convolution launch of ONE D-FINE-s 640x640 bs 16 fp32 train step (BASELINE config #2), replayed alone
from a HIP graph: count per step, us per launch, TFLOP/s, TB/s of compulsory bytes, and the launch's own bound
max(FLOPs / 157 TFLOP/s (f32 MFMA), bytes / 8 TB/s).  The launches are recorded at hip.py's wrappers during one eager step; the
first call of every signature keeps its tensors alive.
GPU box only:   python tools/f32_table.py [--model s --img 640 --batch 16] [--full 40]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd import hip
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s")
ap.add_argument("--img", type=int, default=640)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--full", type=int, default=60, help="rows printed")
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step(a.model, a.img, dev, None)
step.hip_graph = False
images, targets = make_batch(a.batch, a.img, seed=42, device=dev)
for _ in range(3):
    step(images, list(targets))
torch.cuda.synchronize()

groups = collections.OrderedDict()
names = ("gemm_f32_nt", "gemm_f32", "conv1x1_f32", "conv_f32_forward", "conv_f32_wgrad")
orig = {n: getattr(hip, n) for n in names}
depth = [0]


def rec(key, fl, io, fn, args, kw):
    ent = groups.get(key)
    if ent is None:
        groups[key] = [1, fl, io, fn, args, kw]
    else:
        ent[0] += 1


def w_nt(a_, b_, bias=None, alpha=1.0, act=0, splits=1, out=None):
    M, K = a_.shape[-2], a_.shape[-1]
    N = b_.shape[-2]
    batch = a_.numel() // (M * K)
    shared = b_.dim() == 2
    if depth[0] == 0:
        rec(("nt", batch, M, N, K, splits, "sharedB" if shared else "", act), 2.0 * batch * M * N * K,
            4.0 * (batch * M * K + (1 if shared else batch) * N * K + batch * max(splits, 1) * M * N), orig["gemm_f32_nt"],
            (a_, b_), dict(bias=bias, alpha=alpha, act=act, splits=splits))
    return orig["gemm_f32_nt"](a_, b_, bias=bias, alpha=alpha, act=act, splits=splits, out=out)


def w_g(a_, b_, a_kmajor=False, b_kmajor=False, bias=None, alpha=1.0, act=0, splits=1):
    if a_kmajor:
        K, M = a_.shape[-2], a_.shape[-1]
    else:
        M, K = a_.shape[-2], a_.shape[-1]
    N = b_.shape[-1] if b_kmajor else b_.shape[-2]
    batch = a_.numel() // (M * K)
    shared = b_.dim() == 2
    rec(("g" + ("T" if a_kmajor else "N") + ("N" if b_kmajor else "T"), batch, M, N, K, splits, "sharedB" if shared else "", act),
        2.0 * batch * M * N * K, 4.0 * (batch * M * K + (1 if shared else batch) * N * K + batch * max(splits, 1) * M * N),
        orig["gemm_f32"], (a_, b_), dict(a_kmajor=a_kmajor, b_kmajor=b_kmajor, bias=bias, alpha=alpha, act=act, splits=splits))
    return orig["gemm_f32"](a_, b_, a_kmajor=a_kmajor, b_kmajor=b_kmajor, bias=bias, alpha=alpha, act=act, splits=splits)


def w_c1(x, w2d):
    B, cin, H, W = x.shape
    cout = w2d.shape[0]
    rec(("conv1x1", B, cout, H * W, cin, 1, "", 0), 2.0 * B * H * W * cin * cout, 4.0 * B * H * W * (cin + cout), orig["conv1x1_f32"],
        (x, w2d), {})
    return orig["conv1x1_f32"](x, w2d)


def w_cf(x, w2, cout, ks, stride, pt, pl, out_hw):
    B, cin, hi, wi = x.shape
    rec((f"conv{ks}x{ks}s{stride}", B, cout, out_hw[0] * out_hw[1], cin, 1, "", 0), 2.0 * B * out_hw[0] * out_hw[1] * cin * cout * ks * ks,
        4.0 * B * (hi * wi * cin + out_hw[0] * out_hw[1] * cout), orig["conv_f32_forward"], (x, w2, cout, ks, stride, pt, pl, out_hw), {})
    return orig["conv_f32_forward"](x, w2, cout, ks, stride, pt, pl, out_hw)


def w_cw(x, dy, ks, stride, pt, pl, partials=False):
    B, cin, hi, wi = x.shape
    _, cout, ho, wo = dy.shape
    rec((f"wgrad{ks}x{ks}s{stride}", B, cout, ho * wo, cin, 1, "", 0), 2.0 * B * ho * wo * cin * cout * ks * ks,
        4.0 * B * (hi * wi * cin + ho * wo * cout), orig["conv_f32_wgrad"], (x, dy, ks, stride, pt, pl), dict(partials=partials))
    depth[0] += 1
    try:
        return orig["conv_f32_wgrad"](x, dy, ks, stride, pt, pl, partials=partials)
    finally:
        depth[0] -= 1


hip.gemm_f32_nt, hip.gemm_f32, hip.conv1x1_f32, hip.conv_f32_forward, hip.conv_f32_wgrad = w_nt, w_g, w_c1, w_cf, w_cw
step(images, list(targets))
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(hip, n, f)

stream = torch.cuda.Stream(device=dev)
rows = []
for d, (count, fl, io, fn, args, kw) in groups.items():
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn(*args, **kw)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
            for _ in range(a.reps):
                fn(*args, **kw)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (4 * a.reps) * 1e3
    bound = max(fl / 157e12, io / 8e12) * 1e6
    rows.append((count, us, d, fl, io, bound))
    del g

tot = sum(n * us for n, us, *_ in rows)
totb = sum(r[0] * r[5] for r in rows)
print(f"{'entry':14s} {'batch':>5s} {'M':>6s} {'N':>6s} {'K':>6s} {'spl':>4s} {'':8s} act  n/step   us/launch  us/step  TFLOP/s   TB/s  bound us  bound_frac")
for n, us, d, fl, io, bound in sorted(rows, key=lambda r: -r[0] * r[1])[:a.full]:
    print(f"{d[0]:14s} {d[1]:5d} {d[2]:6d} {d[3]:6d} {d[4]:6d} {d[5]:4d} {d[6]:8s} {d[7]:3d}  {n:5d}  {us:9.1f}  {n * us:8.1f}  {fl / us / 1e6:7.1f}  {io / us / 1e6:5.2f}  {bound:8.1f}  {bound / us:6.3f}")
print(f"sum per step {tot:.0f} us over {sum(r[0] for r in rows)} launches ({len(rows)} shapes), bound {totb:.0f} us, bound_frac {totb / tot:.3f}  (alone, graph replay)")
by = collections.defaultdict(lambda: [0.0, 0, 0.0])
for n, us, d, fl, io, bound in rows:
    e = by[d[0]]
    e[0] += n * us; e[1] += n; e[2] += n * bound
for k, (t, n, b) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:14s} {t:9.0f} us/step  {n:4d} launches  bound {b:8.0f} us  frac {b / t:.3f}")
