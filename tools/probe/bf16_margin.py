"""How close the per-block bf16 criterion of tests/test_model_gpu.py::test_bf16_hip_blocks_no_worse_than_aten_bf16_blocks_m320 comes
to its limit, four repetitions: the largest h / (1.5 a + 2e-4) over blocks and quantities (h = HIP-bf16 distance to fp32, a = ATen-bf16's,
now the largest of three ATen runs).  The HIP side repeats exactly, ATen's moves by 20 % from run to run.  GPU box."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tests.test_model_gpu import bf16_block_parity_table
for rep in range(4):
    t = bf16_block_parity_table(torch.device("cuda", 0))
    worst = []
    for n, r in t.items():
        for qi, q in enumerate(("y", "dx", "dparam")):
            h, a = r["hip"][qi], r["aten"][qi]
            worst.append((h / (1.5 * a + 2e-4), n, q, h, a, r["hip"][3] if q == "dparam" else ""))
    worst.sort(reverse=True)
    print(rep, [(round(w[0], 3), w[1], w[2], f"{w[3]:.2e}", f"{w[4]:.2e}", w[5]) for w in worst[:4]])
