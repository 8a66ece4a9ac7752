"""Kernel durations of the attention forward / backward at the decoder's shape (B 32, H 8, L 492, masked) and the encoder's (L 400)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
for B, L, masked in [(32, 492, True), (32, 400, False)]:
    if os.environ.get("ONLY_MASKED") == "1" and not masked:
        continue
    H, E = 8, 256
    q, k, v, do = (torch.randn(B, L, E, device=dev).bfloat16() for _ in range(4))
    mask = None
    if masked:
        m = torch.zeros(L, L, dtype=torch.bool, device=dev)
        m[192:, :192] = True
        for g in range(0, 192, 32):
            m[g:g + 32, :g] = True; m[g:g + 32, g + 32:192] = True
        mask = m.view(torch.uint8)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    def run():
        o, lse = hip.attn_forward(q, k, v, H, mask)
        hip.attn_backward(q, k, v, o, do, lse, H, dq, dk, dv, mask)
    for _ in range(3): run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): run()
        torch.cuda.synchronize()
    print(f"B {B} L {L} masked {masked}")
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:4]:
        print(f"   {e.device_time_total / e.count:8.1f} us x{e.count // 5}  {e.key[:70]}")
