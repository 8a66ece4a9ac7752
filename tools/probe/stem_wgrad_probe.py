"""Stand-alone stem weight-gradient launches at the D-FINE-m bs=32 shapes (kernel durations from the profiler)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from custom_d_fine_amd import hip
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
cases = [("stem1 3->24 3x3 s2", 3, 24, 640, 3, 2, 1), ("stem2a 24->12 2x2", 24, 12, 320, 2, 1, 0), ("stem2b 12->24 2x2", 12, 24, 320, 2, 1, 0),
         ("stem3 48->24 3x3 s2", 48, 24, 320, 3, 2, 1), ("stem4 24->32 1x1", 24, 32, 160, 1, 1, 0)]
only = os.environ.get("ONLY")
for name, cin, cout, hin, ks, s, pad in cases:
    if only and only not in name:
        continue
    ho = hin // s
    x = torch.randn(32, cin, hin, hin, device=dev).bfloat16()
    dy = torch.randn(32, cout, ho, ho, device=dev).bfloat16()
    for _ in range(2): hip.stem_wgrad(x, dy, ks, s, pad)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(4): hip.stem_wgrad(x, dy, ks, s, pad)
        torch.cuda.synchronize()
    t = {e.key[:44]: e.device_time_total / e.count for e in prof.key_averages() if e.device_time_total > 0}
    gf = 2.0 * 32 * ho * ho * cin * cout * ks * ks / 1e9
    mb = 2.0 * 32 * (hin * hin * cin + ho * ho * cout) / 1e6
    print(f"{name:22s} {gf:6.1f} GFLOP {mb:6.0f} MB  " + ", ".join(f"{k.replace('void dfine::', '')} {v:.1f} us" for k, v in t.items()))
