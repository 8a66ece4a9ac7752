"""Per-stream device timeline of one steady-state train step (torch.profiler chrome trace): busy time of every HIP stream, the
union, the idle time of the main chain, and a 1-ms histogram of the occupancy of the two streams - where the step is bound by
the device chain, where by the host (the main stream idles while the host is still enqueueing) and where the side stream
(weight gradients) is the longer one.   GPU box only:   python tools/stream_timeline.py [--ms 1.0]"""
import argparse, json, os, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--ms", type=float, default=1.0)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--graph", type=int, default=0, help="1: backbone + encoder through the captured HIP graphs")
ap.add_argument("--cuda-only", type=int, default=0, help="1: device activities only (the CPU-side tracer slows the host: the eager decoder stretch then looks host-bound)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
step.hip_graph = bool(a.graph)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(a.warmup):
    step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA] if a.cuda_only else [ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(images, targets)
    if a.cuda_only:                  # three steps without a sync in between: the middle one runs with the host already ahead, as in steady state
        step(images, targets)
        step(images, targets)
    torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), "trace.json")
prof.export_chrome_trace(path)
ev = [e for e in json.load(open(path))["traceEvents"]
      if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
ev.sort(key=lambda e: e["ts"])
if a.cuda_only:
    marks = [e["ts"] for e in ev if "stem_conv_s2_vec" in e["name"]]          # the first convolution of a step's forward pass
    if len(marks) >= 3:
        ev = [e for e in ev if marks[1] <= e["ts"] < marks[2]]
t0 = ev[0]["ts"]
t1 = max(e["ts"] + e["dur"] for e in ev)
streams = collections.defaultdict(list)
for e in ev:
    streams[e["args"].get("stream", -1)].append((e["ts"] - t0, e["ts"] - t0 + e["dur"], e["name"]))


def union(iv):
    tot, end = 0.0, -1.0
    for s, e, _ in sorted(iv):
        if e > end:
            tot += e - max(s, end)
            end = e
    return tot


order = sorted(streams, key=lambda s: -union(streams[s]))
main = order[0]
print(f"device span of the step {(t1 - t0) / 1e3:.2f} ms, union busy {union([x for s in streams.values() for x in s]) / 1e3:.2f} ms")
for s in order:
    iv = streams[s]
    print(f"  stream {s}: {len(iv):5d} events, busy {union(iv) / 1e3:7.2f} ms, first {iv[0][0] / 1e3:6.2f} ms, last {max(e for _, e, _ in iv) / 1e3:6.2f} ms")
nb = int((t1 - t0) / 1e3 / a.ms) + 1
occ = {s: [0.0] * nb for s in order[:3]}
for s in order[:3]:
    for b, e, _ in streams[s]:
        i = int(b / 1e3 / a.ms)
        while b < e and i < nb:
            hi = min(e, (i + 1) * a.ms * 1e3)
            occ[s][i] += hi - b
            b, i = hi, i + 1
print(f"\noccupancy per {a.ms} ms bin (percent busy): bin start | " + " | ".join(f"stream {s}" for s in order[:3]) + " | longest kernel of the main stream in the bin")
for i in range(nb):
    lo, hi = i * a.ms * 1e3, (i + 1) * a.ms * 1e3
    names = [(e - b, n) for b, e, n in streams[main] if b < hi and e > lo]
    top = max(names)[1][:60] if names else "-"
    print(f"{i * a.ms:6.1f} | " + " | ".join(f"{100 * occ[s][i] / (a.ms * 1e3):5.0f}" for s in order[:3]) + f" | {top}")
# idle windows of the main stream (> 20 us) and what the host was doing
gaps, end = [], 0.0
for b, e, n in sorted(streams[main]):
    if b - end > 20 and end > 0:
        gaps.append((b - end, end, n))
    end = max(end, e)
print(f"\nmain stream idle windows > 20 us: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e3:.2f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:8.1f} us at t = {g[1] / 1e3:6.2f} ms, before {g[2][:80]}")
# what the other streams run while the main stream sits in its longest idle window (the tail the step waits for)
if gaps and len(order) > 1:
    g = max(gaps)
    lo, hi = g[1], g[1] + g[0]
    print(f"\nother streams during the longest main idle window ({lo / 1e3:.2f} - {hi / 1e3:.2f} ms):")
    for s in order[1:3]:
        for b, e, n in sorted(streams[s]):
            if b < hi and e > lo:
                print(f"  stream {s} {b / 1e3:7.3f} + {e - b:7.1f} us  {n[:100]}")

if os.environ.get("TIMELINE_MAIN_TOP"):
    # where the MAIN stream's time goes (it is the step's critical path: 96-100 % busy), by kernel and by phase (forward graph =
    # before the first side-stream launch minus the decoder / criterion, which starts at the first kernel of the matcher)
    import re
    agg = collections.defaultdict(lambda: [0.0, 0])
    for b, e, n in streams[main]:
        k = re.sub(r"\(.*", "", n)[:90]
        agg[k][0] += e - b
        agg[k][1] += 1
    print(f"\nmain stream by kernel (top {os.environ['TIMELINE_MAIN_TOP']}):")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ["TIMELINE_MAIN_TOP"])]:
        print(f"{t / 1e3:7.3f} ms {c:4d} x  {k}")

if os.environ.get("TIMELINE_WINDOW"):
    # kernels of the main stream inside a time window "a,b" (ms from the step's first kernel): e.g. the decoder + criterion stretch
    wa, wb = (float(v) * 1e3 for v in os.environ["TIMELINE_WINDOW"].split(","))
    import re
    agg = collections.defaultdict(lambda: [0.0, 0])
    busy = 0.0
    for b, e, n in streams[main]:
        if b >= wa and e <= wb:
            k = re.sub(r"\(.*", "", n)[:100]
            agg[k][0] += e - b
            agg[k][1] += 1
            busy += e - b
    print(f"\nmain stream inside {wa / 1e3:.1f} .. {wb / 1e3:.1f} ms: busy {busy / 1e3:.2f} ms, {sum(c for _, c in agg.values())} launches")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TIMELINE_WINDOW_TOP", "40"))]:
        print(f"{t / 1e3:7.3f} ms {c:4d} x  {k}")

if os.environ.get("TIMELINE_AROUND"):
    # every launch whose name contains the pattern, with the launches in front of it on ALL streams that overlap it
    pat = os.environ["TIMELINE_AROUND"]
    allev = sorted(((b, e, n, s) for s, iv in streams.items() for b, e, n in iv))
    for b, e, n, s in allev:
        if pat in n:
            print(f"\n{n[:70]}  stream {s}  start {b / 1e3:8.3f} ms  dur {e - b:7.1f} us")
            for b2, e2, n2, s2 in allev:
                if (b2, e2, n2, s2) != (b, e, n, s) and b2 < e and e2 > b:
                    print(f"      overlaps: stream {s2} {b2 / 1e3:8.3f} + {e2 - b2:7.1f} us  {n2[:80]}")

if os.environ.get("TIMELINE_LONGEST"):
    # the longest launches of the main stream with what ran beside them (pathological sharing shows up here: a latency-bound
    # kernel of 256 fat workgroups behind a grouped weight-gradient launch that holds every CU)
    allev = sorted(((b, e, n, s) for s, iv in streams.items() for b, e, n in iv))
    for b, e, n in sorted(streams[main], key=lambda x: x[0] - x[1])[:int(os.environ["TIMELINE_LONGEST"])]:
        print(f"\n{e - b:7.1f} us at {b / 1e3:7.3f} ms  {n[:90]}")
        for b2, e2, n2, s2 in allev:
            if s2 != main and b2 < e and e2 > b:
                print(f"      beside: stream {s2} {b2 / 1e3:8.3f} + {e2 - b2:7.1f} us  {n2[:80]}")

if os.environ.get("TIMELINE_SEQ"):
    # the main stream's launches inside a time window "a,b" (ms) in order
    wa, wb = (float(v) * 1e3 for v in os.environ["TIMELINE_SEQ"].split(","))
    print(f"\nmain stream, {wa / 1e3:.2f} .. {wb / 1e3:.2f} ms in order:")
    for b, e, n in streams[main]:
        if b >= wa and b <= wb:
            print(f"  {b / 1e3:8.3f} + {e - b:7.1f} us  {n[:150]}")

if os.environ.get("TIMELINE_HIST"):
    # launches whose name contains the pattern: count and time per millisecond of the step, all streams
    pat = os.environ["TIMELINE_HIST"]
    hist = collections.defaultdict(lambda: [0, 0.0])
    for s, iv in streams.items():
        for b, e, n in iv:
            if pat in n:
                hist[(s, int(b // 1000))][0] += 1
                hist[(s, int(b // 1000))][1] += e - b
    print(f"\n'{pat}' per ms of the step (stream, ms): count, us")
    for k in sorted(hist):
        print(f"  stream {k[0]}  {k[1]:3d} ms  {hist[k][0]:4d}  {hist[k][1]:7.1f} us")
