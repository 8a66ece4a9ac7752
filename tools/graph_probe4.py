"""GraphedSegment (explicit capture) probes (GPU box; each case in a subprocess)."""
import os, subprocess, sys
CASES = [("conv2d_miopen", {}), ("hip_dw_bn_units", {}), ("hip_mfma_conv_units", {"DFINE_CONV_TUNE": "hip"}),
         ("backbone_aten", {"DFINE_HIP_UNITS": "0"}), ("backbone_hip", {}), ("encoder_aten", {"DFINE_HIP_UNITS": "0"}),
         ("encoder_hip", {}), ("attention_only", {})]
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.nn as nn
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import ConvBNAct, LightConvBNAct
    from custom_d_fine_amd.d_fine.arch.hybrid_encoder import TransformerEncoderLayer
    from custom_d_fine_amd.dl.engine import GraphedSegment
    dev = torch.device("cuda", 0)
    case = sys.argv[2]
    class T(nn.Module):
        def __init__(s, m): super().__init__(); s.m = m
        def forward(s, *a):
            o = s.m(*a) if len(a) == 1 else s.m(list(a))
            return tuple(o) if isinstance(o, (list, tuple)) else (o,)
    def go(mod, inp):
        mod = T(mod).to(dev).train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(2):
                outs = mod(*inp); sum(o.float().sum() for o in outs).backward()
        torch.cuda.synchronize()
        g = GraphedSegment(mod, inp, amp_dtype=torch.bfloat16)
        for _ in range(3):
            outs = g(*inp); sum(o.float().sum() for o in outs).backward()
        torch.cuda.synchronize()
        print("OK", float(outs[0].float().abs().mean()))
    x64 = torch.randn(8, 64, 40, 40, device=dev, requires_grad=True)
    if case == "conv2d_miopen": go(nn.Conv2d(64, 64, 3, padding=1, bias=False), (x64,))
    elif case == "hip_dw_bn_units": go(nn.Sequential(ConvBNAct(64, 64, 3, groups=64, use_lab=True), ConvBNAct(64, 64, 5, groups=64, use_lab=True)), (x64,))
    elif case == "hip_mfma_conv_units": go(nn.Sequential(LightConvBNAct(64, 64, 5, use_lab=True), ConvBNAct(64, 64, 3, use_lab=True)), (x64,))
    elif case == "attention_only": go(TransformerEncoderLayer(256, 8, 1024, 0.0, "gelu"), (torch.randn(4, 400, 256, device=dev, requires_grad=True),))
    else:
        m = dfine.build_model("s", 80, False, "cuda", img_size=[640, 640]).train()
        x = torch.rand(4, 3, 640, 640, device=dev)
        if case.startswith("backbone"): go(m.backbone, (x,))
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats = tuple(f.detach().float().requires_grad_(True) for f in m.backbone(x))
            go(m.encoder, feats)
else:
    for name, env in CASES:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", name], env=e, capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l and "AccumulateGrad" not in l and "run_backward" not in l]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:220] for t in lines[-3:])}", flush=True)
