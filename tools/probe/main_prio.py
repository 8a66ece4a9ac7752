"""Does a high-priority MAIN stream (torch priority -1) change the step time?  (side stream priority: SIDE_PRIO -> hip._SIDE_PRIORITY)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from custom_d_fine_amd import hip
hip._SIDE_PRIORITY = int(os.environ.get("SIDE_PRIO", "0"))      # read when the side stream is first made
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
prio = int(os.environ.get("MAIN_PRIO", "0"))
st = torch.cuda.Stream(device=dev, priority=prio)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
with torch.cuda.stream(st):
    for _ in range(10):
        step(images, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step(images, targets)
    torch.cuda.synchronize()
    print(f"main prio {prio} side prio {hip._SIDE_PRIORITY}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step")
