"""Prints the per-block bf16 parity table the test `test_bf16_hip_blocks_no_worse_than_aten_bf16_blocks_m320` asserts on:
distance 1 - cos to the block's fp32 result of (a) the HIP bf16 path, (b) the ATen bf16 composition."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import bf16_block_parity_table

t = bf16_block_parity_table(torch.device("cuda", 0))
print(f"{'block':36s} {'y hip':>9s} {'y aten':>9s} {'dx hip':>9s} {'dx aten':>9s} {'dp hip':>9s} {'dp aten':>9s}  worst parameter (hip)")
for n, r in t.items():
    h, a = r["hip"], r["aten"]
    print(f"{n:36s} {h[0]:9.2e} {a[0]:9.2e} {h[1]:9.2e} {a[1]:9.2e} {h[2]:9.2e} {a[2]:9.2e}  {h[3]}")
