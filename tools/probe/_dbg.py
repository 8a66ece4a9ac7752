import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.export_program import export_program, load_program
from tests import helpers
m = dfine.build_model("n", 80, False, "cpu", img_size=[320, 320])
m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
m = m.to("cuda")
path, ep = export_program(m, 80, (320, 320), "/tmp/model.pt2", batch=2, half=True)
c = collections.Counter(str(n.target) for n in ep.graph.nodes if n.op == "call_function")
print(c.most_common(40))
print(sum(c.values()), "nodes", os.path.getsize(path) >> 20, "MiB")
for mod in ep.graph_module.modules():
    g = getattr(mod, "graph", None)
    if g is None: continue
    for n in g.nodes:
        if n.op == "call_function" and "aten.linear" in str(n.target):
            print("LINEAR", [getattr(a, "meta", {}).get("val", None).shape if hasattr(a, "meta") and a.meta.get("val") is not None else a for a in n.args])
            print((n.meta.get("stack_trace") or "")[-900:])
