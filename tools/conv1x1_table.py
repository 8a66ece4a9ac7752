"""Every 1x1-convolution launch (forward + data gradient) of ONE D-FINE-m 640x640 bs 32 train step, replayed alone from a HIP
graph (20 launches per replay: no host cost per launch): count per step, us per launch, TFLOP/s, TB/s of compulsory bytes and
the launch's own roofline bound max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s) - the per-shape table behind DESIGN.md section 5.
The launches are recorded at hip.py's launch wrappers during one eager step; the first call of every shape keeps its tensors.
GPU box only:   python tools/conv1x1_table.py [--ks 1] [--model m --img 640 --batch 32]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd import hip
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="m")
ap.add_argument("--img", type=int, default=640)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ks", type=int, default=1)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step(a.model, a.img, dev, torch.bfloat16)
step.hip_graph = False
images, targets = make_batch(a.batch, a.img, seed=42, device=dev)
for _ in range(3):
    step(images, list(targets))
torch.cuda.synchronize()

# the four launch wrappers of hip.py that reach the dense 1x1 / 3x3 forward kernels (forward and data gradient); the FIRST call
# of every signature keeps its tensors alive, so that the replay below runs on the step's own buffers
groups = collections.OrderedDict()


def sig_plain(name, x, cout, ks, accum):
    B, cin, H, W = x.shape
    return (name, B, cin, cout, H * W, ks, "", accum)


def rec(key, fn, args):
    ent = groups.get(key)
    if ent is None:
        groups[key] = [1, fn, args]
    else:
        ent[0] += 1


orig = {n: getattr(hip, n) for n in ("conv_forward_bf16", "conv1x1_accumulate", "conv_accumulate_bf16", "conv1x1_seg_forward")}


def w_fwd(x, w2, cout, ks):
    rec(sig_plain("conv_fwd", x, cout, ks, 0), orig["conv_forward_bf16"], (x, w2, cout, ks))
    return orig["conv_forward_bf16"](x, w2, cout, ks)


def w_acc1(x, w2, y):
    rec(sig_plain("conv1x1_accum", x, y.shape[1], 1, 1), orig["conv1x1_accumulate"], (x, w2, y))
    return orig["conv1x1_accumulate"](x, w2, y)


def w_acc(x, w2, y, ks):
    rec(sig_plain("conv_accum", x, y.shape[1], ks, 1), orig["conv_accumulate_bf16"], (x, w2, y, ks))
    return orig["conv_accumulate_bf16"](x, w2, y, ks)


def w_seg(x_parts, w2, y_parts, accum=False):
    B, _, H, W = x_parts[0].shape
    cin, cout = sum(t.shape[1] for t in x_parts), sum(t.shape[1] for t in y_parts)
    if accum in (False, True):
        extra, tag = (cout if accum else 0), ("+acc" if accum else "")
    else:
        extra = sum(t.shape[1] for t, f in zip(y_parts, accum) if f)
        tag = "+acc" + "".join("1" if f else "0" for f in accum)
    key = ("conv1x1_seg", B, cin, cout, H * W, 1, f" {len(x_parts)}->{len(y_parts)}{tag}", extra / max(cout, 1))
    rec(key, orig["conv1x1_seg_forward"], (list(x_parts), w2, list(y_parts), accum))
    return orig["conv1x1_seg_forward"](x_parts, w2, y_parts, accum)


hip.conv_forward_bf16, hip.conv1x1_accumulate, hip.conv_accumulate_bf16, hip.conv1x1_seg_forward = w_fwd, w_acc1, w_acc, w_seg
step(images, list(targets))
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(hip, n, f)

stream = torch.cuda.Stream(device=dev)
rows = []
for d, (count, fn, args) in groups.items():
    if d[5] != a.ks:
        continue
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn(*args)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
            for _ in range(a.reps):
                fn(*args)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * a.reps) * 1e3
    _, B, cin, cout, HW, KS, parts, acc_frac = d
    fl = 2.0 * B * HW * cin * cout * KS * KS
    io = 2.0 * B * HW * (cin + cout * (1 + acc_frac)) + 2.0 * cin * cout * KS * KS
    bound = max(fl / 2.5e15, io / 8e12) * 1e6
    rows.append((count, us, d, fl, io, bound))
    del g

tot = sum(n * us for n, us, *_ in rows)
totb = sum(r[0] * r[5] for r in rows)
print(f"{'entry':34s} {'B':>3s} {'Cin':>5s} {'Cout':>5s} {'HW':>6s}  n/step   us/launch  us/step  TFLOP/s   TB/s  bound us  bound_frac")
for n, us, d, fl, io, bound in sorted(rows, key=lambda r: -r[0] * r[1]):
    print(f"{d[0] + d[6]:34s} {d[1]:3d} {d[2]:5d} {d[3]:5d} {d[4]:6d}  {n:5d}  {us:9.1f}  {n * us:8.1f}  {fl / us / 1e6:7.1f}  {io / us / 1e6:5.2f}  {bound:8.1f}  {bound / us:6.3f}")
print(f"sum per step {tot:.0f} us over {sum(r[0] for r in rows)} launches, bound {totb:.0f} us, bound_frac {totb / tot:.3f}  (ks = {a.ks}, alone, graph replay)")
