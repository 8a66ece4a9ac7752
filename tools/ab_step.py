"""A/B timing of one switch inside ONE process: the train step alternates between the two settings in blocks of a few steps
(ABAB...), so box-to-box and minute-to-minute drift (+-1 ms between two bench.py runs on these boxes) cancels.  GPU box:
    python tools/ab_step.py hip.WGRAD_STREAM            # module attribute toggled False / True
    python tools/ab_step.py env:DFINE_GRAD_FANIN        # environment switch "0" / "1" + kernels.reload_env()
    python tools/ab_step.py hip._SIDE_GROUP_AT=12,48    # module attribute set to the first / second value
    python tools/ab_step.py step.hip_graph              # eager / HIP-graph replay of backbone + encoder
    AB_BLOCKS=12 AB_STEPS=8 python tools/ab_step.py ..."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd import hip, kernels
from custom_d_fine_amd.dl.synthetic import make_batch

what = sys.argv[1]
blocks, steps = int(os.environ.get("AB_BLOCKS", "10")), int(os.environ.get("AB_STEPS", "8"))
dev = torch.device("cuda", 0)
step = bench.build_step(os.environ.get("SP_MODEL", "m"), int(os.environ.get("SP_IMG", "640")), dev, torch.bfloat16)
images, targets = make_batch(int(os.environ.get("SP_BATCH", "32")), int(os.environ.get("SP_IMG", "640")), seed=42, device=dev)


_HI = torch.cuda.Stream(device=dev, priority=-1)
_CTX = [None]


def setting(on):
    if what == "main_high_priority":                   # the whole step on a high-priority stream (the side stream stays normal)
        torch.cuda.synchronize()
        if _CTX[0] is not None:
            _CTX[0].__exit__(None, None, None)
            _CTX[0] = None
        if on:
            _CTX[0] = torch.cuda.stream(_HI)
            _CTX[0].__enter__()
        return
    if what == "step.hip_graph":                       # backbone + encoder through the captured HIP graphs (dl/engine.GraphedSegment)
        step.hip_graph = bool(on)
        return
    if what.startswith("env:"):
        os.environ[what[4:]] = "1" if on else "0"
        kernels.reload_env()
    elif "=" in what:                                  # hip._SIDE_GROUP_AT=12,48 : arm 0 -> 12, arm 1 -> 48
        name, vals = what.split("=")
        mod, attr = name.split(".")
        a, b = vals.split(",")
        from custom_d_fine_amd.dl import engine
        mods = {"hip": hip, "kernels": kernels, "seg": engine.GraphedSegment}        # seg.GROUP_AT=8,16 (class attributes; AB_RECAPTURE=1)
        old = getattr(mods[mod], attr)
        val = tuple(filter(None, (b if on else a).split("+"))) if isinstance(old, tuple) else type(old)(b if on else a)   # tuples: "c3+k2"
        setattr(mods[mod], attr, val)
        if os.environ.get("AB_RECAPTURE") == "1":      # the setting is baked into the captured graphs: build them again
            torch.cuda.synchronize()
            for seg in list(step._graphs.values()):
                seg.release()
            step._graphs.clear()
        if attr == "_SIDE_PRIORITY":                   # the side stream is made again with the new priority
            torch.cuda.synchronize()
            hip._SIDE.clear()
    else:
        mod, attr = what.split(".")
        setattr({"hip": hip, "kernels": kernels}[mod], attr, bool(on))


for on in (False, True):
    setting(on)
    for _ in range(4):
        step(images, targets)
torch.cuda.synchronize()
times = {False: [], True: []}
for blk in range(2 * blocks):
    on = bool(blk & 1)
    setting(on)
    step(images, targets)                       # one untimed step after the switch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(images, targets)
    torch.cuda.synchronize()
    times[on].append((time.perf_counter() - t0) * 1e3 / steps)
for on in (False, True):
    v = times[on]
    print(f"{what} = {int(on)}: median {statistics.median(v):.3f} ms/step  mean {statistics.fmean(v):.3f}  min {min(v):.3f}  ({len(v)} blocks of {steps} steps)")
print(f"difference of the medians (1 - 0): {statistics.median(times[True]) - statistics.median(times[False]):+.3f} ms")
