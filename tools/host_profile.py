"""Host-side (Python) time of steady-state train steps: cProfile over a few steps, top functions by
cumulative and by own time.  The GPU queue hides host time only while the host runs ahead; phases where
it does not (criterion, start of backward) show up as device idle gaps in tools/gpu_gaps.py.
GPU box only:   python tools/host_profile.py [--steps 3]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--rows", type=int, default=45)
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(a.warmup):
    step(images, targets)
torch.cuda.synchronize()
# un-profiled: how long does the host wait for the device (matcher sync + final sync)?  step - wait = host work
_orig_sync = torch.cuda.Event.synchronize
_wait = [0.0]
def _timed_sync(self):
    t = time.perf_counter(); _orig_sync(self); _wait[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = _timed_sync
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n):
    step(images, targets)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
torch.cuda.Event.synchronize = _orig_sync
print(f"un-profiled: {1e3*(t2-t0)/n:.1f} ms/step wall; host waits at the matcher sync {1e3*_wait[0]/n:.1f} ms/step, final drain {1e3*(t2-t1):.1f} ms"
      f" -> host work ~{1e3*((t1-t0)-_wait[0])/n:.1f} ms/step")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(a.steps):
    step(images, targets)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time {1e3*(t1-t0)/a.steps:.1f} ms/step (profiled), drain {1e3*(t2-t1):.1f} ms")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(a.rows)
st.sort_stats("tottime").print_stats(a.rows)
