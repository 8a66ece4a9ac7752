"""Attention kernels stand-alone at the decoder's real shapes and mask (300 queries + denoising groups) - per-kernel device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
B, H, E, NQ = 32, 8, 256, 300
for maxT, groups in ((15, 6), (9, 11), (0, 0)):
    dn = 2 * maxT * groups
    L = NQ + dn
    mask = torch.zeros(L, L, dtype=torch.bool, device=dev)
    if dn:
        mask[dn:, :dn] = True
        for i in range(groups):
            a, b = 2 * maxT * i, 2 * maxT * (i + 1)
            mask[a:b, :a] = True
            mask[a:b, b:dn] = True
    for name, m in (("real", mask), ("real-nosum", mask), ("random", (torch.rand(L, L, device=dev) < 0.3).fill_diagonal_(False)), ("none", None)):
        hip._MASK_SUMMARY = name != "real-nosum"
        hip._MASK_BITS[0] = None
        m8 = None if m is None else m.view(torch.uint8).contiguous()
        qkv = torch.randn(B, L, 3 * E, device=dev).bfloat16()
        q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
        do = torch.randn(B, L, E, device=dev).bfloat16()
        o, lse2 = hip.attn_forward(q, k, v, H, m8)
        dqkv = torch.empty_like(qkv)
        f = lambda: hip.attn_backward(q, k, v, o, do, lse2, H, dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:], m8)
        for _ in range(3): f(); hip.attn_forward(q, k, v, H, m8)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(5): hip.attn_forward(q, k, v, H, m8); f()
            torch.cuda.synchronize()
        row = {k_.key.split("<")[0].split("::")[-1]: k_.device_time_total / 5 for k_ in prof.key_averages() if "attn" in k_.key}
        print(f"L {L} mask {name:10s}: " + "  ".join(f"{a} {b:7.1f} us" for a, b in sorted(row.items())))
