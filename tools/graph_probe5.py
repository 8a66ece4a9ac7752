"""Which ingredient of GraphedSegment crashes?  MLP only (GPU box; each case in a subprocess)."""
import os, subprocess, sys
CASES = ["base", "autocast", "allow_unused", "warm_grad", "segment_noamp", "segment_amp", "segment_conv_noamp"]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.nn as nn
    dev = torch.device("cuda", 0)
    case = sys.argv[2]
    mod = nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256)).to(dev)
    x = torch.randn(64, 256, device=dev, requires_grad=True)
    params = list(mod.parameters())
    if case.startswith("segment"):
        from custom_d_fine_amd.dl.engine import GraphedSegment
        class T(nn.Module):
            def __init__(s, m): super().__init__(); s.m = m
            def forward(s, a): return (s.m(a),)
        if case == "segment_conv_noamp":
            mod = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev); x = torch.randn(8, 64, 40, 40, device=dev, requires_grad=True)
        g = GraphedSegment(T(mod), (x,), amp_dtype=torch.bfloat16 if case == "segment_amp" else None)
        for _ in range(3):
            o = g(x); o[0].float().sum().backward()
        torch.cuda.synchronize(); print("OK", float(o[0].float().abs().mean())); sys.exit(0)
    amp = case in ("autocast",)
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp, cache_enabled=False):
            return mod(x)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y = run()
            if case == "warm_grad": torch.autograd.grad((y,), [x] + params, (torch.ones_like(y),), allow_unused=True)
            else:
                y.sum().backward()
                for p in params: p.grad = None
                x.grad = None
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        y = run()
    gy = torch.ones_like(y)
    with torch.cuda.graph(g2, pool=g1.pool()):
        grads = torch.autograd.grad((y,), [x] + params, (gy,), allow_unused=case in ("allow_unused", "warm_grad"))
    g1.replay(); g2.replay(); torch.cuda.synchronize()
    print("OK", float(grads[1].float().abs().mean()))
else:
    for name in CASES:
        r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l and "AccumulateGrad" not in l and "run_backward" not in l]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:220] for t in lines[-3:])}", flush=True)
