"""Lists every host<->device synchronisation inside one steady-state train step
(torch.cuda.set_sync_debug_mode("warn")) with the Python line that caused it.  The step is meant to have
exactly one: the matcher's D2H copy of the assignment.  GPU box only:   python tools/sync_check.py"""
import os, sys, traceback, warnings
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
seen = []


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    st = [f for f in traceback.extract_stack() if "/custom_d_fine_amd/" in f.filename or f.filename.endswith("bench.py")]
    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-4:]))
    if not where:       # raised from ATen itself (e.g. inside the autograd engine): keep the whole Python stack
        where = "[no package frame] " + " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(traceback.extract_stack()[-8:-1]))
    seen.append(where + "   | " + str(message)[:80])


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(images, targets)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"{len(seen)} synchronising calls in one train step:")
for w in seen:
    print("  ", w)
