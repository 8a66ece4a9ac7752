"""ATen ops of one EAGER train step that touch a tensor of >= 16 M elements (name, shapes, dtypes, call site) - GPU box."""
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
        ts = [a for a in args if torch.is_tensor(a)]
        if ts and max(t.numel() for t in ts) >= (1 << 24) and func.__name__.split(".")[0] not in ("view", "reshape", "detach", "_unsafe_view", "alias", "t", "transpose", "permute", "slice", "select", "unsqueeze", "expand", "as_strided", "split_with_sizes", "unbind", "squeeze"):
            fr = [f for f in traceback.extract_stack() if "custom_d_fine_amd" in f.filename and "probe" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(engine)"
            sites[(func.__name__, tuple((tuple(t.shape), str(t.dtype).split(".")[1]) for t in ts[:3]), where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(images, list(targets))
for k, n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(n, k)
