"""Manual autograd-under-capture probes (GPU box; each case in a subprocess)."""
import os, subprocess, sys
CASES = ["whole_fwd_bwd", "two_graph_manual", "two_graph_no_pool"]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import torch, torch.nn as nn
    dev = torch.device("cuda", 0)
    case = sys.argv[2]
    mod = nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256)).to(dev)
    x = torch.randn(64, 256, device=dev, requires_grad=True)
    params = list(mod.parameters())
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y = mod(x); y.sum().backward()
            for p in params: p.grad = None
            x.grad = None
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    if case == "whole_fwd_bwd":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = mod(x); y.sum().backward()
        g.replay(); torch.cuda.synchronize()
        print("OK", float(params[0].grad.abs().mean()))
    else:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            y = mod(x)
        gy = torch.ones_like(y)
        if case == "two_graph_manual":
            with torch.cuda.graph(g2, pool=g1.pool()):
                grads = torch.autograd.grad((y,), [x] + params, (gy,), only_inputs=True, allow_unused=False)
        else:
            with torch.cuda.graph(g2):
                grads = torch.autograd.grad((y,), [x] + params, (gy,))
        g1.replay(); g2.replay(); torch.cuda.synchronize()
        print("OK", float(grads[1].abs().mean()))
else:
    for name in CASES:
        r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:200] for t in lines[-3:])}", flush=True)
