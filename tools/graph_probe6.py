"""GraphedSegment on nn.Conv2d with single extra ingredients (GPU box; each case in a subprocess)."""
import os, subprocess, sys
CASES = ["plain", "import_hip", "eager_backward_first", "amp", "amp_eager", "train_to", "dfine_import"]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.nn as nn
    case = sys.argv[2]
    if case == "import_hip": import custom_d_fine_amd.hip
    if case == "dfine_import": from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import GraphedSegment
    dev = torch.device("cuda", 0)
    class T(nn.Module):
        def __init__(s, m): super().__init__(); s.m = m
        def forward(s, a): return (s.m(a),)
    mod = T(nn.Conv2d(64, 64, 3, padding=1, bias=False)).to(dev)
    if case == "train_to": mod = mod.train()
    x = torch.randn(8, 64, 40, 40, device=dev, requires_grad=True)
    amp = torch.bfloat16 if case.startswith("amp") else None
    if case in ("eager_backward_first", "amp_eager"):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            for _ in range(2):
                o = mod(x); o[0].float().sum().backward()
        torch.cuda.synchronize()
    g = GraphedSegment(mod, (x,), amp_dtype=amp)
    for _ in range(3):
        o = g(x); o[0].float().sum().backward()
    torch.cuda.synchronize(); print("OK", float(o[0].detach().float().abs().mean()))
else:
    for name in CASES:
        r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l and "AccumulateGrad" not in l and "run_backward" not in l]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:220] for t in lines[-2:])}", flush=True)
