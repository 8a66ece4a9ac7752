"""Margins of tests/test_train_trace.py::test_train_trace_gpu per parameter (cosine and step-length ratio of the 3-iteration weight
delta against the golden trace), with and without the fp32 column-sum kernel.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_train_trace as T, helpers
from custom_d_fine_amd import hip

def margins(tag):
    model, ema, rec = T._run(torch.device("cuda", 0), fused_opt=True)
    G = T.G
    sd = model.state_dict()
    init = helpers.seeded_state_dict(sd)
    rows = []
    for key in [k for k in G.files if k.startswith("final/") and k != "final/num_batches_tracked"]:
        name = key.split("/", 1)[1]
        want, got, w0 = torch.tensor(G[key]), sd[name].detach().cpu().float(), init[name].float()
        dw, dg = (want - w0).flatten(), (got - w0).flatten()
        cos = torch.nn.functional.cosine_similarity(dw, dg, dim=0).item()
        rows.append((name, cos, dw.norm().item() / dg.norm().item(), dw.numel()))
    print(tag, "losses", [round(r[1], 4) for r in rec])
    for r in sorted(rows, key=lambda r: r[1])[:6]:
        print(tag, "cos  ", r)
    for r in sorted(rows, key=lambda r: -abs(r[2] - 1))[:6]:
        print(tag, "ratio", r)

margins("colsum on ")
hip.colsum_f32_ok = lambda d: False
margins("colsum off")
