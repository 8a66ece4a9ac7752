"""Wall time of the train-step phases with a device sync after each (GPU box)."""
import os, sys, time
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
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0) + (t - t0); return t
N = 5
for _ in range(N):
    t = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = step.model(images, targets=targets)
    t = tick("forward", t)
    heads = [{k: v for k, v in out.items() if "aux" not in k}] + list(out["aux_outputs"]) + [out["pre_outputs"]] + list(out["enc_aux_outputs"])
    m = step.criterion.matcher.match_heads(heads, targets)
    t = tick("matcher(extra, also inside criterion)", t)
    loss_dict = step.criterion(out, targets)
    loss = sum(loss_dict.values())
    t = tick("criterion", t)
    loss.backward()
    t = tick("backward", t)
    step.optimizer_step()
    t = tick("clip+adamw+ema", t)
for k, v in acc.items():
    print(f"{k:45s} {v / N * 1e3:8.2f} ms")
# host-only cost of the criterion: run it again without syncs and time the python side
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    loss_dict = step.criterion(out, targets)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"criterion host issue time {(t1 - t0) / N * 1e3:.2f} ms, + drain {(t2 - t1) / N * 1e3:.2f} ms")
