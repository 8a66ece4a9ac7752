"""fp32 <-> bf16 casts, cats and strided copies of one train step in the eager decoder / criterion stretch, by call site (GPU box)."""
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
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in ("_to_copy", "copy_", "cat", "clone", "contiguous", "add", "stack"):
            ts = [a for a in (args[0] if name in ("cat", "stack") else args) if torch.is_tensor(a)]
            if ts and ts[0].is_cuda and ts[0].numel() >= 100000:
                fr = [f for f in traceback.extract_stack() if "custom_d_fine_amd" in f.filename and "probe" not in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:]) if fr else "(engine)"
                dt = "/".join(str(t.dtype).split(".")[1][:4] for t in ts[:2])
                to = str((kwargs or {}).get("dtype", "")).split(".")[-1][:4]
                sites[(name, tuple(ts[0].shape), dt, to, ts[0].is_contiguous(), where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(images, list(targets))
for k, n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print(n, k)
