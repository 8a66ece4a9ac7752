"""Python call sites of host->device transfers (torch.tensor(..., device=cuda), .to(cuda), .cuda(), copy_ from a CPU tensor)
in one steady-state train step (GPU box).  Each is a small blit kernel on the device and ~10 us of host time."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
sites = collections.Counter()
active = [False]

def site():
    for f in reversed(traceback.extract_stack(limit=12)[:-2]):
        if "custom_d_fine_amd" in f.filename or f.filename.endswith("bench.py"):
            return f"{f.filename.split('custom_d_fine_amd/')[-1]}:{f.lineno} {f.line[:70]}"
    return "?"

def is_cuda_dev(d):
    try:
        return d is not None and torch.device(d).type == "cuda"
    except Exception:
        return False

def wrap_factory(name):
    orig = getattr(torch, name)
    def f(*a, **k):
        if active[0] and is_cuda_dev(k.get("device")):
            sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(torch, name, f)
for n in ("tensor", "as_tensor", "full", "scalar_tensor"):
    if n in ("tensor", "as_tensor", "scalar_tensor"):
        wrap_factory(n)

orig_to = torch.Tensor.to
def to(self, *a, **k):
    if active[0] and not self.is_cuda:
        tgt = k.get("device", a[0] if a else None)
        if isinstance(tgt, torch.Tensor):
            tgt = tgt.device
        if is_cuda_dev(tgt) if not isinstance(tgt, torch.dtype) else False:
            sites[("to", site())] += 1
    return orig_to(self, *a, **k)
torch.Tensor.to = to
orig_cuda = torch.Tensor.cuda
def cuda(self, *a, **k):
    if active[0] and not self.is_cuda:
        sites[("cuda", site())] += 1
    return orig_cuda(self, *a, **k)
torch.Tensor.cuda = cuda
orig_copy = torch.Tensor.copy_
def copy_(self, src, *a, **k):
    if active[0] and self.is_cuda and isinstance(src, torch.Tensor) and not src.is_cuda:
        sites[("copy_", site())] += 1
    return orig_copy(self, src, *a, **k)
torch.Tensor.copy_ = copy_

active[0] = True
step(images, targets)
torch.cuda.synchronize()
active[0] = False
for (kind, s), n in sites.most_common(60):
    print(f"{n:4d} {kind:10s} {s}")
print("total", sum(sites.values()))
