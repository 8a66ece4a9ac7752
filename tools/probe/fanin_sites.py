"""Gradient fan-in sites of the autograd graph of one eager train step: nodes whose output gradient arrives over >= 2 edges (autograd adds
them with an element-wise kernel), with the node that produced the map and the nodes that consume it - the candidates for
accumulate-into epilogues (GPU box)."""
import os, sys, collections
os.environ["DFINE_HIPGRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
model, crit = step.model, step.criterion
model.train()
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = model(images, targets)
loss = sum(crit(out, targets).values())
indeg, consumers, shape = collections.Counter(), collections.defaultdict(list), {}
seen, stack, keep = set(), [loss.grad_fn], []          # (node wrappers are kept alive: their ids are the keys)
while stack:
    n = stack.pop()
    if n is None or id(n) in seen:
        continue
    seen.add(id(n))
    keep.append(n)
    for nxt, idx in n.next_functions:
        if nxt is None:
            continue
        keep.append(nxt)
        indeg[(id(nxt), idx)] += 1
        consumers[(id(nxt), idx)].append(type(n).__name__)
        shape[(id(nxt), idx)] = (type(nxt).__name__, getattr(nxt, "_input_metadata", None))
        stack.append(nxt)
rows = collections.Counter()
for key, d in indeg.items():
    if d < 2:
        continue
    name, _ = shape[key]
    if name == "AccumulateGrad":
        continue
    rows[(name, key[1], tuple(sorted(consumers[key])))] += 1
for (name, idx, cons), n in sorted(rows.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:3d} x output {idx} of {name:28s} <- {', '.join(cons)}")
