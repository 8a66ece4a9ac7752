"""The element-wise adds the autograd engine itself issues in one EAGER train step (gradient fan-in of maps with several consumers),
by shape and bytes moved - what the GradFanIn hand-offs have not removed yet (GPU box; encoder-output parking is capture-only)."""
import os, sys, collections, traceback
os.environ["DFINE_HIPGRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(3):
    step(images, list(targets))
sites = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in ("add", "add_"):
            ts = [a for a in args if torch.is_tensor(a)]
            fr = [f for f in traceback.extract_stack() if "custom_d_fine_amd" in f.filename and "probe" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(engine)"
            sites[(name, tuple(ts[0].shape), str(ts[0].dtype).split(".")[1], where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(images, list(targets))
def nbytes(k):
    n = 1
    for d in k[1]:
        n *= d
    return n * (2 if k[2] == "bfloat16" else 4) * 3
tot = 0
for k, n in sorted(sites.items(), key=lambda kv: -kv[1] * nbytes(kv[0])):
    mb = n * nbytes(k) / 1e6
    tot += mb
    if mb > 1:
        print(f"{n:4d} x {k[0]:5s} {str(k[1]):26s} {k[2]:9s} {mb:8.1f} MB  {k[3]}")
print("total MB", tot)
