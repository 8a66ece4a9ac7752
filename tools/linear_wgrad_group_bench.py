"""The grouped linear weight-gradient launch (dfine_linear_wgrad_group) on a decoder step's problem list (D-FINE-m: 4 layers x
16 linears over M = 15 744 token rows), timed from a HIP graph; A/B of two builds through DFINE_HIP_LIB.
GPU box only:   python tools/linear_wgrad_group_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from custom_d_fine_amd import hip
from custom_d_fine_amd.d_fine.arch.utils import upload

dev = torch.device("cuda", 0)
M = 15744
LAYER = [(768, 256), (256, 256), (192, 256), (96, 256), (256, 256), (1024, 256), (256, 1024), (512, 512), (80, 256), (256, 256), (256, 256),
         (132, 256), (64, 20), (1, 64), (512, 4), (256, 256)]
probs = LAYER * 4
torch.manual_seed(0)
xs = {k: torch.randn(M, k, device=dev).bfloat16() for k in {p[1] for p in probs}}
dys = {n: torch.randn(M, n, device=dev).bfloat16() for n in {p[0] for p in probs}}
table = np.empty((len(probs), 8), dtype=np.int64)
keep, blocks, flops = [], 1, 0.0
for i, (N, K) in enumerate(probs):
    ws = torch.empty(int(hip._lib.dfine_linear_wgrad_ws_floats(M, N, K)), device=dev)
    keep.append(ws)
    n = int(hip._lib.dfine_linear_wgrad_group_row(xs[K].data_ptr(), dys[N].data_ptr(), ws.data_ptr(), M, N, K, table[i].ctypes.data))
    blocks = max(blocks, n)
    flops += 2.0 * M * N * K
dev_table = upload(table, dev)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    for _ in range(2):
        hip._check(hip._lib.dfine_linear_wgrad_group(dev_table.data_ptr(), len(probs), blocks, st.cuda_stream), "group")
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
        for _ in range(5):
            hip._check(hip._lib.dfine_linear_wgrad_group(dev_table.data_ptr(), len(probs), blocks, st.cuda_stream), "group")
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
# check one problem against fp32 torch
N, K = probs[5]
splits = int(hip._lib.dfine_linear_wgrad_splits(M, N, K))
np16, cp16 = (N + 15) // 16 * 16, (K + 15) // 16 * 16
part = keep[5][:splits * np16 * cp16].view(splits, np16, cp16).sum(0)[:N, :K]
want = dys[N].float().t() @ xs[K].float()
err = ((part - want).abs().max() / want.abs().max()).item()
print(f"{os.environ.get('DFINE_HIP_LIB', 'tree build'):40s} {len(probs)} problems, grid x {blocks}: {us:8.1f} us per launch, {flops / us / 1e6:6.1f} TFLOP/s, rel err {err:.1e}")
