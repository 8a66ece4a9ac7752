"""Ordered device events of ONE steady-state train step in the timed mode (graph replay of backbone + encoder, two streams):
torch.profiler with device activities only, three steps without a sync in between, the middle one written as JSON lines
[start_us, dur_us, stream, name] - the raw material for per-stretch analyses (decoder + criterion stretch, launch gaps).
GPU box only:   python tools/step_events.py OUT.jsonl [--model m --img 640 --batch 32]"""
import argparse, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--model", default="m")
ap.add_argument("--img", type=int, default=640)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--warmup", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
step = bench.build_step(a.model, a.img, dev, torch.bfloat16)
images, targets = make_batch(a.batch, a.img, seed=42, device=dev)
for _ in range(a.warmup):
    step(images, list(targets))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step(images, list(targets))
    torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), "trace.json")
prof.export_chrome_trace(path)
ev = [e for e in json.load(open(path))["traceEvents"] if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
ev.sort(key=lambda e: e["ts"])
marks = [e["ts"] + e["dur"] for e in ev if "sqnorm_final_kernel" in e["name"]]      # one per step, in the optimizer tail
if len(marks) >= 3:
    ev = [e for e in ev if marks[0] <= e["ts"] < marks[1]]          # (starts with the tail of step 1's optimizer, ends before step 2's)
t0 = ev[0]["ts"]
with open(a.out, "w") as f:
    for e in ev:
        f.write(json.dumps([round(e["ts"] - t0, 2), round(e["dur"], 2), e["args"].get("stream", -1), e["name"][:200]]) + "\n")
print(f"{len(ev)} events, span {(max(e['ts'] + e['dur'] for e in ev) - t0) / 1e3:.2f} ms -> {a.out}")
