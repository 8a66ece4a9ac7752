"""Host-side cost of the FORWARD pass only (the host-bound phase of the step): cProfile over a few forward passes of the
bench model under autocast, top functions by own time.  GPU box only:   python tools/host_forward_profile.py"""
import cProfile, os, pstats, sys, time
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
model = step.model


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model(images, targets=targets)


for _ in range(2):
    fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = fwd()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"forward: host enqueue {1e3 * (t1 - t0) / 5:.1f} ms per pass, drain {1e3 * (t2 - t1):.1f} ms after 5 passes")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    out = fwd()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
