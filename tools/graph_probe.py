"""Which part of backbone/encoder survives HIP-graph capture?  (GPU box; each case in a subprocess)"""
import os, subprocess, sys
CASES = [
    ("backbone, all HIP", {"PART": "backbone"}),
    ("backbone, MFMA conv off", {"PART": "backbone", "DFINE_MFMA_CONV": "0"}),
    ("backbone, pure ATen units", {"PART": "backbone", "DFINE_HIP_UNITS": "0"}),
    ("encoder, pure ATen units", {"PART": "encoder", "DFINE_HIP_UNITS": "0"}),
    ("encoder, all HIP", {"PART": "encoder"}),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from custom_d_fine_amd.d_fine import dfine
    dev = torch.device("cuda", 0)
    m = dfine.build_model("s", 80, False, "cuda", img_size=[640, 640]).train()
    x = torch.rand(4, 3, 640, 640, device=dev)
    part = os.environ["PART"]
    if part == "backbone":
        mod, inp = m.backbone, (x,)
        wrap = lambda f: (lambda *a: tuple(f(*a)))
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats = [f.detach().requires_grad_(True) for f in m.backbone(x)]
        class E(torch.nn.Module):
            def __init__(s, e): super().__init__(); s.e = e
            def forward(s, a, b, c): return tuple(s.e([a, b, c]))
        mod, inp = E(m.encoder), tuple(feats)
    class B(torch.nn.Module):
        def __init__(s, b): super().__init__(); s.b = b
        def forward(s, a): return tuple(s.b(a))
    if part == "backbone": mod = B(m.backbone)
    with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
        for _ in range(2):
            out = mod(*inp); sum(o.float().sum() for o in out).backward()
        torch.cuda.synchronize()
        g = torch.cuda.make_graphed_callables(mod, inp, num_warmup_iters=2)
        for _ in range(3):
            out = g(*inp); sum(o.float().sum() for o in out).backward()
        torch.cuda.synchronize()
    print("OK", float(out[0].float().abs().mean()))
else:
    for name, env in CASES:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
        print(f"[{name}] rc={r.returncode} :: {' | '.join(t[:160] for t in tail)}", flush=True)
