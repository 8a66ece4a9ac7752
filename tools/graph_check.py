"""HIP-graph replay of backbone+encoder must reproduce the eager losses (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
from custom_d_fine_amd.d_fine.arch import utils as U
dev = torch.device("cuda", 0)
res = {}
for flag in ("0", "1"):
    os.environ["DFINE_HIPGRAPH"] = flag
    torch.manual_seed(0)
    step = bench.build_step("s", 640, dev, torch.bfloat16)
    images, targets = make_batch(4, 640, seed=1, device=dev)
    out = []
    for it in range(6):
        U.set_denoising_generator(torch.Generator().manual_seed(100 + it))
        loss, _ = step(images, targets)
        out.append(loss.item())
    U.set_denoising_generator(None)
    res[flag] = out
    print(flag, ["%.4f" % v for v in out])
worst = max(abs(a - b) / abs(a) for a, b in zip(res["0"], res["1"]))
print("worst relative loss difference eager vs graph:", worst)
assert worst < 2e-2
