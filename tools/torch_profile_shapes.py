"""Which tensors do the remaining ATen elementwise ops (copy_/add/cat/sum/mul/fill_) touch? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4): step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): step(images, targets)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::mul", "aten::fill_", "aten::mm", "aten::addmm", "aten::bmm",
                 "aten::native_layer_norm", "aten::native_layer_norm_backward", "aten::_softmax", "aten::gelu", "aten::relu", "aten::sigmoid", "aten::upsample_nearest2d", "aten::clamp", "aten::index", "aten::gather", "aten::zeros"):
        rows.append((e.self_device_time_total / 2e3, e.count // 2, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for r in rows[:70]:
    print(f"{r[0]:7.3f} ms/step x{r[1]:4d} {r[2]:28s} {r[3]}")
