import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep
from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(100)
model = dfine.build_model("n", 5, False, "cuda:0", img_size=[320, 320]).train()
crit = dfine.build_loss("n", 5, 0.0, False)
ema = ModelEMA(model, 0.9998)
opt = dfine.build_optimizer(model, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
fused = FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1, overlap=True, bucket_mb=2)
names = {id(p): n for n, p in model.named_parameters()}
pname = [names[id(p)] for p in fused._params]
import collections
calls = collections.Counter()
for i, p_ in enumerate(fused._params):
    def mk(i):
        def h(_p):
            calls[("hook", i)] += 1
        return h
    p_.register_post_accumulate_grad_hook(mk(i))
orig_ready = fused.param_ready
def param_ready(index):
    calls[("ready", index)] += 1
    b = fused._buckets[fused._bucket_of[index]]
    if b["done"]:
        pass
    orig_ready(index)
fused.param_ready = param_ready
orig_reduce = fused._reduce_bucket
def reduce_bucket(b):
    if b["ready"] != len(b["params"]):
        missing = [pname[i] for i in b["params"] if fused._params[i].grad is None][:6]
        print("bucket flushed at collect: ready", b["ready"], "of", len(b["params"]), "no-grad params e.g.", missing)
    orig_reduce(b)
fused._reduce_bucket = reduce_bucket
step = TrainStep(model, crit, opt, amp_dtype=torch.bfloat16, clip_max_norm=0.1, ema=ema, fused_optimizer=fused)
images, targets = make_batch(2, 320, num_classes=5, seed=42, device=dev)
for it in range(1):
    print("step", it)
    step(images, targets)
per = collections.Counter()
for (kind, i), n in calls.items():
    per[i] += n
print("params with != 1 notifications:", [(pname[i], [(k, n) for (k, j), n in calls.items() if j == i]) for i in range(len(pname)) if per[i] != 1][:40])
print("buckets:", [(b["lo"], b["hi"], len(b["params"])) for b in fused._buckets])
torch.cuda.synchronize()
print("uses left:", {pname[k]: v for k, v in fused._uses.items() if v != 0})
