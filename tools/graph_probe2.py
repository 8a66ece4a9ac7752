"""Finer HIP-graph capture probe (GPU box; each case in a subprocess)."""
import os, subprocess, sys
CASES = ["fwd_only_backbone", "conv2d_miopen", "hip_units_only", "light_unit_hip_conv", "linear_only", "conv2d_miopen_fp32"]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.nn as nn
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import ConvBNAct, LightConvBNAct
    dev = torch.device("cuda", 0)
    case = sys.argv[2]
    def run_gc(mod, inp, amp=True):
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False, enabled=amp):
            for _ in range(2):
                out = mod(*inp); out.float().sum().backward()
            torch.cuda.synchronize()
            g = torch.cuda.make_graphed_callables(mod, inp, num_warmup_iters=2)
            for _ in range(3):
                out = g(*inp); out.float().sum().backward()
            torch.cuda.synchronize()
        return float(out.float().abs().mean())
    if case == "fwd_only_backbone":
        m = dfine.build_model("s", 80, False, "cuda", img_size=[640, 640]).train()
        x = torch.rand(4, 3, 640, 640, device=dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            for _ in range(2): o = m.backbone(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                o = m.backbone(x)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                o = m.backbone(x)
            g.replay(); torch.cuda.synchronize()
        print("OK", float(o[-1].float().abs().mean()))
    elif case == "conv2d_miopen":
        print("OK", run_gc(nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev), (torch.randn(8, 64, 40, 40, device=dev, requires_grad=True),)))
    elif case == "conv2d_miopen_fp32":
        print("OK", run_gc(nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev), (torch.randn(8, 64, 40, 40, device=dev, requires_grad=True),), amp=False))
    elif case == "hip_units_only":
        mod = nn.Sequential(ConvBNAct(64, 64, 3, groups=64, use_lab=True), ConvBNAct(64, 64, 5, groups=64, use_lab=True)).to(dev).train()
        print("OK", run_gc(mod, (torch.randn(8, 64, 40, 40, device=dev, requires_grad=True).bfloat16().detach().requires_grad_(True),)))
    elif case == "light_unit_hip_conv":
        mod = nn.Sequential(LightConvBNAct(64, 64, 5, use_lab=True), ConvBNAct(64, 64, 3, use_lab=True)).to(dev).train()
        print("OK", run_gc(mod, (torch.randn(8, 64, 40, 40, device=dev, requires_grad=True),)))
    elif case == "linear_only":
        print("OK", run_gc(nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256)).to(dev), (torch.randn(64, 256, device=dev, requires_grad=True),)))
else:
    for name in CASES:
        r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l and "AccumulateGrad" not in l and "run_backward" not in l]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:200] for t in lines[-3:])}", flush=True)
