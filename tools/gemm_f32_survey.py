"""Per-shape cost of the fp32 GEMM launches (dfine_gemm_f32 / _nt / _nn) of one D-FINE-s bs=16 fp32 train step (BASELINE
config #2): shapes collected from a real step, each timed stand-alone (kernel durations from the profiler).   GPU box only."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd import hip
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
step = bench.build_step("s", 640, dev, None)
step.hip_graph = False
images, targets = make_batch(16, 640, seed=42, device=dev)
step(images, targets)
seen = collections.OrderedDict()
orig = {n: getattr(hip, n) for n in ("gemm_f32_nt", "gemm_f32", "conv1x1_f32")}


def spy(name):
    def f(*a, **k):
        ts = [t for t in a if torch.is_tensor(t)]
        key = (name, tuple(tuple(t.shape) for t in ts[:2]), tuple(sorted((kk, vv) for kk, vv in k.items() if not torch.is_tensor(vv) and vv is not None)),
               tuple(x for x in a if isinstance(x, (bool, int)) and not torch.is_tensor(x)))
        if key not in seen:
            seen[key] = [0, a, k]
        seen[key][0] += 1
        return orig[name](*a, **k)
    return f


for n in orig:
    setattr(hip, n, spy(n))
step(images, targets)
torch.cuda.synchronize()
for n in orig:
    setattr(hip, n, orig[n])
rows = []
for key, (cnt, a, k) in seen.items():
    name = key[0]
    for _ in range(2):
        orig[name](*a, **k)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            orig[name](*a, **k)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if "gemm_f32" in e.name]
    us = sum(e.device_time for e in evs) / 5
    s0, s1 = key[1][0], key[1][1]
    if name == "conv1x1_f32":
        B, cin, H, W = s0
        fl = 2.0 * B * H * W * cin * s1[0]
    elif name == "gemm_f32_nt":
        batch = 1
        for d in s0[:-2]:
            batch *= d
        fl = 2.0 * batch * s0[-2] * s0[-1] * s1[-2]
    else:
        akm = k.get("a_kmajor", False) or (len(key[3]) > 0 and key[3][0])
        bkm = k.get("b_kmajor", False) or (len(key[3]) > 1 and key[3][1])
        batch = 1
        for d in s0[:-2]:
            batch *= d
        M, K = (s0[-1], s0[-2]) if akm else (s0[-2], s0[-1])
        N = s1[-1] if bkm else s1[-2]
        fl = 2.0 * batch * M * N * K
    rows.append((us * cnt, cnt, us, fl / max(us, 1e-9) / 1e6, key))
rows.sort(key=lambda r: -r[0])
print(f"total {sum(r[0] for r in rows) / 1e3:.2f} ms in {sum(r[1] for r in rows)} calls, {len(rows)} shapes")
for tot, cnt, us, tf, key in rows[:45]:
    print(f"{tot / 1e3:7.2f} ms {cnt:4d} x {us:8.1f} us {tf:7.1f} TFLOP/s  {key[0]} {key[1]} {key[2]} {key[3]}")
