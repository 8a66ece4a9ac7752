"""Python call sites of the ATen element-wise ops left in the eager decoder / criterion stretch (TorchDispatchMode + traceback)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4):
    step(images, list(targets))
sites = collections.Counter()
WANT = ("add", "mul", "copy_", "_to_copy", "cat", "fill_", "zero_", "clone", "sub", "div", "where", "stack", "sum", "neg", "sigmoid", "clamp", "index", "gather", "scatter")
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if any(name.startswith(w) for w in WANT):
            shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), None)
            fr = [f for f in traceback.extract_stack() if "custom_d_fine_amd" in f.filename and "probe" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd engine)"
            sites[(name, shp, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(images, list(targets))
for (name, shp, where), n in sorted(sites.items(), key=lambda kv: -kv[1] * (1 if kv[0][1] is None else max(1, int(torch.tensor(kv[0][1]).prod()))))[:70]:
    print(f"{n:4d} x {name:10s} {str(shp):24s} {where}")
