"""Per-shape time of the token-stream GEMM and attention kernels vs the library path (hipBLASLt addmm / SDPA) at the
shapes of one D-FINE-m bs=32 train step (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from custom_d_fine_amd import hip

dev = torch.device("cuda", 0)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("linear: M K N | hip us (TF/s) | addmm us")
for M, K, N in [] if os.environ.get("LAB_SKIP_LINEAR") == "1" else [(15744, 256, 256), (15744, 256, 512), (15744, 512, 256), (15744, 256, 1024), (15744, 1024, 256), (15744, 512, 512),
                (15744, 256, 192), (15744, 256, 96), (15744, 256, 80), (15744, 256, 132), (15744, 4, 512), (15744, 20, 64), (15744, 64, 1),
                (12800, 256, 256), (12800, 256, 1024), (12800, 1024, 256), (268800, 256, 256), (268800, 256, 80), (9600, 256, 256)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    bb = b.bfloat16()
    th = t(lambda: hip.linear_act(x, w, b))
    ta = t(lambda: torch.addmm(bb, x, w.t()))
    print(f"{M:7d} {K:5d} {N:5d} | {th:8.1f} ({2.0*M*K*N/th/1e6:6.1f}) | {ta:8.1f}")

print("attention: B L masked | fwd us, bwd us | sdpa fwd us, bwd us")
for B, L, masked in [(32, 492, True), (32, 400, False)]:
    H, E = 8, 256
    qk = torch.randn(B, L, 2 * E, device=dev).bfloat16()
    v = torch.randn(B, L, E, device=dev).bfloat16()
    do = torch.randn(B, L, E, device=dev).bfloat16()
    mask = (torch.rand(L, L, device=dev) < 0.3) if masked else None
    if mask is not None:
        mask.fill_diagonal_(False)
    m8 = None if mask is None else mask.view(torch.uint8)
    q, k = qk[..., :E], qk[..., E:]
    o, lse2 = hip.attn_forward(q, k, v, H, m8)
    dqk, dv = torch.empty_like(qk), torch.empty_like(v)
    tf = t(lambda: hip.attn_forward(q, k, v, H, m8))
    tb = t(lambda: hip.attn_backward(q, k, v, o, do, lse2, H, dqk[..., :E], dqk[..., E:], dv, m8))
    qh, kh, vh = (x_.reshape(B, L, H, 32).transpose(1, 2).detach().requires_grad_(True) for x_ in (q, k, v))
    am = None if mask is None else ~mask
    oo = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=am)
    go = torch.randn_like(oo)
    tsf = t(lambda: F.scaled_dot_product_attention(qh, kh, vh, attn_mask=am))
    tsb = t(lambda: torch.autograd.grad(oo, (qh, kh, vh), go, retain_graph=True))
    print(f"{B:3d} {L:4d} {masked!s:5} | {tf:7.1f} {tb:7.1f} | {tsf:7.1f} {tsb:7.1f}")
