"""(f1) Inference table of `Torch_model` on the HIP forward path - the counterpart of the reference's batch-size sweep
(`src/dl/test_batching.py:16-60`: N images per batch size, wall clock around `model(batch)`, README.md:159-171: D-FINE-m
640x640 on an RTX 5070 Ti, 76.4 / 113.4 / 138.1 / 122.7 / 119.7 / 117.8 images/s at bs 1 / 2 / 4 / 8 / 16 / 32) and of its
latency table (`src/dl/bench.py:80-120`: 10 warm-up calls, synchronize around each call, mean of the rest; README.md:111:
16.6 ms end to end at bs 1).  End to end = uint8 HWC frames already on the host -> device pre-processing kernel -> model
-> device post-processing -> per-image result dicts on the host side of the call.

    python tools/infer_bench.py [--model m] [--images 256] [--out profiles/r03_infer_table.txt]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from custom_d_fine_amd.infer.torch_model import Torch_model

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="m")
ap.add_argument("--images", type=int, default=256, help="images per (configuration, batch size) - the reference uses 512")
ap.add_argument("--out", default=None)
ap.add_argument("--graph", type=int, default=0, help="1: also measure Torch_model(hip_graph=True)")
args = ap.parse_args()
assert torch.cuda.is_available()
rng = np.random.default_rng(0)
frames = rng.integers(0, 255, size=(32, 720, 1280, 3), dtype=np.uint8)         # HD frames, resized on the device
REF = {1: 76.4, 2: 113.4, 4: 138.1, 8: 122.7, 16: 119.7, 32: 117.8}
lines = []


def emit(s):
    print(s, flush=True)
    lines.append(s)


emit(f"Torch_model D-FINE-{args.model} 640x640, random-init weights, 80 classes, 1280x720 uint8 source frames, {args.images} images per cell, "
     f"one MI355X; reference column: README.md:159-171 (RTX 5070 Ti, Torch fp32)")
emit(f"{'precision':10s} {'deploy':7s} {'graph':6s} {'bs':>3s} {'ms/batch':>9s} {'ms/img':>8s} {'img/s':>9s} {'ref img/s':>9s}  model-only ms/batch")
for half, deploy, graph in [(h, d, g) for g in ((False, True) if args.graph else (False,)) for h in (False, True) for d in (False, True)]:
    if True:
        tm = Torch_model(args.model, None, 80, 640, 640, half=half, hip_graph=False)
        if deploy:
            tm.model.deploy()
        tm.hip_graph = graph
        for bs in (1, 2, 4, 8, 16, 32):
            batch = frames[:bs] if bs > 1 else frames[0]
            for _ in range(10):
                tm(batch)
            torch.cuda.synchronize()
            n = max(args.images // bs, 4)
            t0 = time.perf_counter()
            for _ in range(n):
                tm(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            # network alone on a device-resident, pre-processed batch
            x, _, _ = tm._prepare_inputs(batch)
            for _ in range(3):
                tm._predict(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(n // 2, 4)):
                tm._predict(x)
            torch.cuda.synchronize()
            dm = (time.perf_counter() - t1) / max(n // 2, 4)
            emit(f"{'bf16' if half else 'fp32':10s} {str(deploy):7s} {str(graph):6s} {bs:3d} {dt * 1e3:9.2f} {dt * 1e3 / bs:8.3f} {bs / dt:9.1f} {REF[bs]:9.1f}  {dm * 1e3:8.2f}")
        del tm
        torch.cuda.empty_cache()
if args.out:
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
