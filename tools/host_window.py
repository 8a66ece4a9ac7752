"""Host work between the matcher sync and the first encoder-backward kernel of one train step - the part of the step in
which the device waits for the host (the queue is empty after the sync; criterion, autograd start-up and the decoder's
backward are many small launches).  Outermost host ops of the window by name: count, total us; then the sequence of the
longest ones.  The profiler slows the host down by ~25 %: read the shares, not the absolute times.
GPU box only:   python tools/host_window.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(5):
    step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(images, targets)
    torch.cuda.synchronize()
evs = list(prof.events())
cpu = sorted((e for e in evs if str(e.device_type).endswith("CPU")), key=lambda e: e.time_range.start)
gpu = sorted((e for e in evs if str(e.device_type).endswith("CUDA")), key=lambda e: e.time_range.start)
sync = next(e for e in cpu if e.name == "hipEventSynchronize")
lo = sync.time_range.end
first_bwd = next(e for e in gpu if e.time_range.start > lo and ("bn2_bwd" in e.name or "bn_bwd" in e.name or "bn_one_bwd" in e.name))
launch = [e for e in cpu if e.name.startswith("hipLaunchKernel") or e.name.startswith("hipExtModuleLaunch")]
# host time at which that kernel was launched: the last launch call that started before the kernel did
hi = max(e.time_range.start for e in launch if e.time_range.start < first_bwd.time_range.start)
t0 = cpu[0].time_range.start
print(f"step host span {(cpu[-1].time_range.end - t0) / 1e3:.2f} ms; matcher sync returns at {(lo - t0) / 1e3:.2f} ms, "
      f"first encoder-backward kernel launched at ~{(hi - t0) / 1e3:.2f} ms (runs at {(first_bwd.time_range.start - t0) / 1e3:.2f} ms): window {(hi - lo) / 1e3:.2f} ms")
busy = sum(min(e.time_range.end, first_bwd.time_range.start) - max(e.time_range.start, lo) for e in gpu
           if e.time_range.end > lo and e.time_range.start < first_bwd.time_range.start)
print(f"device kernel time inside the window (sum over streams): {busy / 1e3:.2f} ms")
top, end = [], -1
for e in cpu:
    if e.time_range.start < lo or e.time_range.start >= hi:
        continue
    if e.time_range.start >= end:
        top.append(e)
        end = e.time_range.end
agg = collections.defaultdict(lambda: [0, 0.0])
for e in top:
    k = e.name.replace("autograd::engine::evaluate_function: ", "bwd ")
    agg[k][0] += 1
    agg[k][1] += e.time_range.end - e.time_range.start
covered = sum(v[1] for v in agg.values())
print(f"outermost host ops: {len(top)} covering {covered / 1e3:.2f} ms of the window (the rest is Python between ops)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {t:9.1f} us  {n:4d} x  {k[:100]}")
print("\nlongest single ops, in order:")
for e in sorted(sorted(top, key=lambda e: -(e.time_range.end - e.time_range.start))[:30], key=lambda e: e.time_range.start):
    print(f"  t = {(e.time_range.start - lo) / 1e3:6.2f} ms  {e.time_range.end - e.time_range.start:8.1f} us  {e.name[:100]}")
print("\nPython-only stretches (no op running) longer than 60 us:")
end = lo
for e in top:
    if e.time_range.start - end > 60:
        print(f"  t = {(end - lo) / 1e3:6.2f} ms  {e.time_range.start - end:8.1f} us  before {e.name[:80]}")
    end = max(end, e.time_range.end)

print("\ndevice kernels inside the window by name (count, total us):")
dagg = collections.defaultdict(lambda: [0, 0.0])
for e in gpu:
    if e.time_range.start >= lo and e.time_range.start < first_bwd.time_range.start:
        dagg[e.name[:110]][0] += 1
        dagg[e.name[:110]][1] += e.time_range.end - e.time_range.start
print(f"  {sum(v[0] for v in dagg.values())} launches")
for k, (n, t) in sorted(dagg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"  {n:4d} x {t:9.1f} us  {k}")
