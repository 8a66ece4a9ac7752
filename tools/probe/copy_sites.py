"""Same-dtype device copies (hipMemcpyAsync D2D = __amd_rocclr_copyBuffer) of one EAGER train step by call site - GPU box."""
import os, sys, collections, traceback
os.environ["DFINE_HIPGRAPH"] = os.environ.get("DFINE_HIPGRAPH", "0")
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
        if name in ("copy_", "clone", "_to_copy", "contiguous"):
            ts = [a for a in args if torch.is_tensor(a)]
            same = len(ts) < 2 or ts[0].dtype == ts[1].dtype
            if name == "_to_copy":
                same = (kwargs or {}).get("dtype", ts[0].dtype) == ts[0].dtype
            if same and ts[0].is_cuda:
                fr = [f for f in traceback.extract_stack() if "custom_d_fine_amd" in f.filename and "probe" not in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:]) if fr else "(engine)"
                sites[(name, tuple(ts[0].shape), str(ts[0].dtype).split(".")[1], where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(images, list(targets))
for k, n in sorted(sites.items(), key=lambda kv: -kv[1])[:45]:
    print(n, k)
print("total", sum(sites.values()))
