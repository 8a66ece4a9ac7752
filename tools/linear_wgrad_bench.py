"""Weight-gradient GEMM of the token-stream linears: dW[N,K] = dY[M,N]^T X[M,K], M = 15744 (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (N, K) in ((256, 256), (1024, 256), (256, 1024), (80, 256), (512, 512), (132, 256), (512, 4), (64, 20), (1, 64)):
    B, L = 32, 492
    x = torch.randn(B, L, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, L, N, device=dev, dtype=torch.bfloat16)
    ref = (dy.reshape(-1, N).float().t() @ x.reshape(-1, K).float())
    a = t(lambda: dy.reshape(-1, N).t() @ x.reshape(-1, K))
    b = t(lambda: torch.bmm(dy.transpose(1, 2), x).sum(0, dtype=torch.float32))
    def hipw():
        xt = x.reshape(-1, K).t().contiguous(); dt = dy.reshape(-1, N).t().contiguous()
        return hip.conv_wgrad_bf16(xt.view(1, K, 1, -1), dt.view(1, N, 1, -1), 1)
    c = t(hipw)
    err = (hipw().view(N, K) - ref).abs().max().item() / ref.abs().max().item()
    # split via mm on chunks stacked (einsum)
    d = t(lambda: hip.linear_wgrad_bf16(x.reshape(-1, K), dy.reshape(-1, N)))
    err2 = (hip.linear_wgrad_bf16(x.reshape(-1, K), dy.reshape(-1, N)) - ref).abs().max().item() / ref.abs().max().item()
    print(f"N={N:4d} K={K:4d}: mm {a:7.1f} us | bmm+sum {b:7.1f} us | transpose+HIP split-K {c:7.1f} us (rel err {err:.1e}) | HIP row-major split-K {d:7.1f} us (rel err {err2:.1e})")

print("forward variants")
import torch.nn.functional as F
for (N, K) in ((1024, 256), (256, 1024), (256, 256)):
    x = torch.randn(32, 492, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(32, 492, N, device=dev, dtype=torch.bfloat16)
    a = t(lambda: F.linear(x, w, b))
    c = t(lambda: torch.addmm(b, x.reshape(-1, K), w.t(), out=out.view(-1, N)))
    d = t(lambda: torch.addmm(b, x.reshape(-1, K), w.t()))
    e = t(lambda: torch.matmul(torch.randn(32, 492, N, device=dev, dtype=torch.bfloat16), w))
    print(f"N={N} K={K}: F.linear {a:.1f} us | addmm(out=) {c:.1f} us | addmm {d:.1f} us | dx matmul(+randn) {e:.1f} us")
