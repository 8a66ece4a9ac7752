"""A few fp32 GEMM shapes of config #2 (1x1 convolutions on NCHW maps, token-stream linears) timed alone from a HIP graph -
for A/B of builds through DFINE_HIP_LIB.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cases = [("conv1x1", 16, 512, 512, 80), ("conv1x1", 16, 256, 256, 80), ("conv1x1", 16, 512, 512, 40), ("conv1x1", 16, 256, 256, 40),
         ("conv1x1", 16, 640, 256, 80), ("conv1x1", 16, 64, 64, 40), ("nt", 7968, 1024, 256, 0), ("nt", 7968, 256, 256, 0), ("nn", 7968, 256, 1024, 0),
         ("tn", 256, 1024, 7968, 31)]
st = torch.cuda.Stream(device=dev)
out = []
for c in cases:
    if c[0] == "conv1x1":
        _, B, cin, cout, H = c
        x = torch.randn(B, cin, H, H, device=dev); w = torch.randn(cout, cin, device=dev)
        fn = lambda: hip.conv1x1_f32(x, w)
        fl = 2.0 * B * H * H * cin * cout
    elif c[0] == "nt":
        _, M, N, K, _ = c
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
        fn = lambda: hip.gemm_f32_nt(a, b)
        fl = 2.0 * M * N * K
    elif c[0] == "nn":
        _, M, N, K, _ = c
        a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
        fn = lambda: hip.gemm_f32(a, b, b_kmajor=True)
        fl = 2.0 * M * N * K
    else:
        _, M, N, K, sp = c
        a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev)
        fn = lambda: hip.gemm_f32(a, b, a_kmajor=True, b_kmajor=True, splits=sp)
        fl = 2.0 * M * N * K
    with torch.cuda.stream(st):
        fn(); fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
            for _ in range(10):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    out.append(f"{str(c):38s} {us:8.1f} us {fl / us / 1e6:6.1f} TF")
print(os.environ.get("DFINE_HIP_LIB", "tree build"))
print("\n".join(out))
