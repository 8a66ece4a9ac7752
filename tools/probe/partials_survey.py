"""Size of the deferred split-K partial-sum buffers of one D-FINE-m bs=32 train step, by weight shape (what multi_wgrad_reduce reads)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
step.hip_graph = False
images, targets = make_batch(32, 640, seed=42, device=dev)
step(images, targets)
f = step.fused
seen = collections.OrderedDict()
orig = f.defer_wgrad
def spy(index, ws, meta, ws_offset=0, dst_offset=0):
    splits, cout, cin, taps, np16, cp16 = meta
    key = (cout, cin, taps)
    e = seen.setdefault(key, [0, 0, 0.0])
    e[0] += 1; e[1] = splits; e[2] += splits * np16 * cp16 * taps * 4 / 1e6
    return orig(index, ws, meta, ws_offset, dst_offset)
f.defer_wgrad = spy
step(images, targets)
torch.cuda.synchronize()
tot = sum(v[2] for v in seen.values())
print(f"deferred partial sums per step: {tot:.0f} MB in {sum(v[0] for v in seen.values())} buffers")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1][2])[:25]:
    print(f"  w [{k[0]:4d}, {k[1]:4d}, {k[2]}]  x{v[0]:3d}  splits {v[1]:4d}  {v[2]:7.1f} MB")
