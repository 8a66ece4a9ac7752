"""Per-shape cost of the 3x3 weight-gradient launches (dfine_conv_wgrad_bf16, ks = 3) of one D-FINE-m bs=32 train step:
shapes collected from a real step, each timed stand-alone with HIP events.   GPU box only."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd import hip
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
step.hip_graph = False
images, targets = make_batch(32, 640, seed=42, device=dev)
step(images, targets)
seen = collections.OrderedDict()
orig = hip.conv_wgrad_bf16


def spy(x, dy, ks, partials=False):
    if ks == 3:
        k = (tuple(x.shape), dy.shape[1])
        seen[k] = seen.get(k, 0) + 1
    return orig(x, dy, ks, partials)


hip.conv_wgrad_bf16 = spy
import custom_d_fine_amd.kernels as K
step(images, targets)
torch.cuda.synchronize()
hip.conv_wgrad_bf16 = orig
tot = 0.0
print(f"{'x shape':>24} {'cout':>5} {'n':>3} {'us':>8} {'TFLOP/s':>8} {'GB/s':>7} {'splits':>6}")
for (xs, cout), n in seen.items():
    x = torch.randn(xs, device=dev).bfloat16()
    dy = torch.randn(xs[0], cout, xs[2], xs[3], device=dev).bfloat16()
    for _ in range(3):
        orig(x, dy, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):                      # (the partial-sum launches go to the side stream: wall clock over a synchronize)
        ws, meta = orig(x, dy, 3, partials=True)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / 50
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            orig(x, dy, 3, partials=True)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if "conv_wgrad_kernel" in e.name or "conv_wgrad3_" in e.name]
    kus = sum(e.device_time for e in evs) / max(1, len(evs))
    others = sorted({e.name[:50] for e in prof.events() if e.device_time > 0 and "conv_wgrad_kernel" not in e.name and "conv_wgrad3_" not in e.name})
    print(f"   kernel alone {kus:7.1f} us ({len(evs)} launches); other device work: {others}")
    us = kus
    fl = 2.0 * xs[0] * xs[2] * xs[3] * xs[1] * cout * 9
    by = 2.0 * xs[0] * xs[2] * xs[3] * (xs[1] + cout)
    tot += us * n
    print(f"{str(xs):>24} {cout:5d} {n:3d} {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:7.0f} {meta[0]:6d}")
print(f"sum over the step: {tot / 1e3:.2f} ms")
