"""linear_act on the token-stream shapes (M = 15 744 decoder rows / 12 800 AIFI rows), isolated, against the library GEMM:
   python tools/probe/linear_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
shapes = [(15744, 256, 256), (15744, 1024, 256), (15744, 256, 1024), (15744, 192, 256), (15744, 96, 256), (15744, 512, 256),
          (12800, 256, 256), (12800, 768, 256), (9600, 256, 256), (268800, 256, 256)]
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_h = timeit(lambda: hip.linear_act(x, w, bias, 0, out=out))
    bb = bias.bfloat16()
    t_l = timeit(lambda: torch.nn.functional.linear(x, w, bb))
    ref = torch.nn.functional.linear(x.float(), w.float(), bias)
    err = (out.float() - ref).abs().max().item()
    io = 2.0 * (M * K + N * K + M * N)
    print(f"M {M:6d} N {N:4d} K {K:4d}: hip {t_h:7.1f} us ({io / t_h / 1e6:5.2f} TB/s, {2.0 * M * N * K / t_h / 1e6:6.1f} TF)  library {t_l:7.1f} us   max err {err:.3f}")
