import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tests.test_model_gpu as T
from custom_d_fine_amd import kernels
cuda = torch.device("cuda:0")
orig = T.bf16_decoder_parity_table
import types
src = open(T.__file__).read()
# run with the key assertion replaced by a print
src = src.replace("            assert gp0.keys() == gp1.keys(), (name, mode, set(gp0) ^ set(gp1))", "            print(name, mode, 'only fp32:', sorted(set(gp0) - set(gp1)), 'only mode:', sorted(set(gp1) - set(gp0)))\n            gp0 = {k: v for k, v in gp0.items() if k in gp1}")
ns = {"__name__": "dbg", "__file__": T.__file__}
exec(compile(src, T.__file__, "exec"), ns)
tab = ns["bf16_decoder_parity_table"](cuda)
for n, row in tab.items():
    print(n, {k: tuple(round(x, 6) if isinstance(x, float) else x for x in v) for k, v in row.items()})
