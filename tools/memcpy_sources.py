"""The small memcpy / memset operations of one train step (torch.profiler chrome trace): kind, size, count - and the host operator that
issued each (by correlation id), to find avoidable copies.   GPU box:   python tools/memcpy_sources.py"""
import json, os, sys, tempfile, collections
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
path = os.path.join(tempfile.mkdtemp(), "trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
gpu = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("gpu_memcpy", "gpu_memset")]
cpu_ops = sorted((e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cpu_op", "user_annotation")), key=lambda e: e["ts"])
rt = {e["args"].get("correlation"): e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cuda_runtime", "cuda_driver") and "args" in e}
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for g in gpu:
    r = rt.get(g["args"].get("correlation"))
    owner = "?"
    if r is not None:
        best = None
        for c in cpu_ops:                       # innermost cpu op that encloses the runtime call
            if c["ts"] <= r["ts"] and c["ts"] + c["dur"] >= r["ts"] + r.get("dur", 0):
                if best is None or c["dur"] < best["dur"]:
                    best = c
            elif c["ts"] > r["ts"]:
                break
        owner = best["name"] if best else "(no cpu op: direct library call)"
    k = (g["name"][:28], int(g["args"].get("bytes", 0)), owner[:70])
    agg[k][0] += 1
    agg[k][1] += g["dur"]
print(f"{len(gpu)} memcpy / memset operations in the step")
for (name, nbytes, owner), (n, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:4d} x {t:8.1f} us  {name:28s} {nbytes:10d} B  {owner}")
