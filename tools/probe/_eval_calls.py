import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.export import DFINEPostProcessor, ExportWrapper

class Proxy:
    def __init__(self, lib): self._lib, self.calls = lib, collections.Counter()
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        def call(*a):
            self.calls[name] += 1
            return fn(*a)
        return call
for size, half in (("m", True), ("m", False), ("x", True)):
    m = dfine.build_model(size, 80, False, "cuda", img_size=[640, 640]).eval()
    m.deploy()
    w = ExportWrapper(m, DFINEPostProcessor(80), (640, 640))
    x = torch.rand(2, 3, 640, 640, device="cuda")
    p = Proxy(hip._lib.__dict__.get("_lib", hip._lib) if isinstance(hip._lib, Proxy) else hip._lib)
    hip._lib = p
    hip._PURE.__dict__.clear()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=half):
        out = w(x)
    hip._lib = p._lib
    print(size, "half" if half else "fp32", dict(p.calls))
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=half):
            out = w(x)
        torch.cuda.synchronize()
    aten = collections.Counter()
    for e in prof.key_averages():
        if "dfine::" not in e.key: aten[e.key[:60]] += e.count
    print("   non-dfine kernels:", dict(aten))
