"""RCCL smoke on a one-GPU box: the data-parallel code path (rank-0 broadcast, bucketed asynchronous all-reduce of flat
gradient slices launched from backward, gloo side group for the criterion's normalisers) with a REAL nccl(=RCCL) process
group of one rank - the modules are told the world has two ranks so that every collective is issued.  Checks that the
collectives run and that the step equals the single-process step (a one-rank all-reduce is the identity; the 1/2 gradient
scale is undone by doubling nothing - Adam is scale-invariant up to eps, so only finiteness and progress are asserted).
    python tools/probe/rccl_one_rank.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="env://", device_id=dev)

from custom_d_fine_amd.d_fine import dfine, dfine_criterion, dist_utils
from custom_d_fine_amd.dl import fused_optim
from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep
from custom_d_fine_amd.dl.synthetic import make_batch

for mod in (fused_optim, dfine_criterion, dist_utils):
    mod.get_world_size = lambda: 2                      # issue every collective although the group has one rank
calls = {"all_reduce": 0, "broadcast": 0}
_ar, _bc = dist.all_reduce, dist.broadcast


def all_reduce(t, *a, **k):
    calls["all_reduce"] += 1
    return _ar(t, *a, **k)


def broadcast(t, *a, **k):
    calls["broadcast"] += 1
    return _bc(t, *a, **k)


dist.all_reduce, dist.broadcast = all_reduce, broadcast
torch.manual_seed(0)
model = dfine.build_model("n", 5, False, "cuda:0", img_size=[320, 320]).train()
crit = dfine.build_loss("n", 5, 0.0, False)
ema = ModelEMA(model, 0.9998)
opt = dfine.build_optimizer(model, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
fused = fused_optim.FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1, bucket_mb=2)
assert fused.overlap and len(fused._buckets) > 4
fused.broadcast_from_rank0()
GRAPH = os.environ.get("RCCL_PROBE_GRAPH", "1") == "1"          # the bench's default mode: captured backbone + encoder segment
step = TrainStep(model, crit, opt, amp_dtype=torch.bfloat16, clip_max_norm=0.1, ema=ema, fused_optimizer=fused, hip_graph=GRAPH)
images, targets = make_batch(4, 320, num_classes=5, seed=42, device=dev)
before = fused.flat_param.clone()
losses = []
order = []
for it in range(4):
    if GRAPH and it == 3 and step._graphs:
        # last step: where do the all-reduces fall relative to the backbone's backward replay?
        chain = next(iter(step._graphs.values()))
        if hasattr(chain, "segments"):
            bb = chain.segments[0]
            gm, gs = bb.bwd_pairs[0]

            class _Spy:
                def replay(self_):
                    order.append("backbone backward replay starts")
                    if os.environ.get("RCCL_PROBE_DEBUG") == "1":
                        names = {id(p): n for n, p in model.named_parameters()}
                        for bi, b in enumerate(fused._buckets):
                            own = sorted({names[id(fused._params[i])].split(".")[0] for i in b["params"]})
                            print(f"  bucket {bi}: {own} params {len(b['params'])} ready {b['ready']} gathered {b['gathered']} done {b['done']} "
                                  f"MB {(b['hi'] - b['lo']) * 4 / 1e6:.2f}", flush=True)
                        print("  next_launch", fused._next_launch, flush=True)
                    gm.replay()
            bb.bwd_pairs[0] = (_Spy(), gs)
            _ar2 = dist.all_reduce

            def all_reduce2(t, *a, **k):
                if t.numel() > 1024:
                    order.append(f"all_reduce of {t.numel() * 4 / 1e6:.2f} MB")
                return _ar2(t, *a, **k)
            dist.all_reduce = all_reduce2
    loss, _ = step(images, targets)
    losses.append(float(loss))
if order:
    print("order of the last step:", order)
    first = order.index("backbone backward replay starts")
    assert first > 0, "no gradient bucket was all-reduced before the backbone's backward started"
torch.cuda.synchronize()
assert all(l == l and abs(l) < 1e6 for l in losses), losses
assert (fused.flat_param - before).abs().max() > 0
assert torch.isfinite(fused.flat_param).all()
assert not GRAPH or step._graphs, "the graph-replay path did not run"
print("backend", dist.get_backend(), "| graph replay", GRAPH, "| collectives issued:", calls, "| losses", [round(l, 3) for l in losses])
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank smoke ok")
