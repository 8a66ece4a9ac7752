"""How much does the data-parallel MODE of the step cost by itself?  The bench step (D-FINE-m 640x640 bs 32 bf16) on a real
nccl(=RCCL) group of ONE rank with the modules told the world has two (bucketed asynchronous all-reduces from backward, per-bucket
deferred reductions, gloo side group) against the plain single-process step on the same box.
    python tools/probe/ddp_mode_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

images, targets = make_batch(32, 640, seed=42, device=dev)


def timed(step, n=30, warm=8):
    for _ in range(warm):
        step(images, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(images, targets)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


single = timed(bench.build_step("m", 640, dev, torch.bfloat16))
dist.init_process_group("nccl", init_method="env://", device_id=dev)
from custom_d_fine_amd.d_fine import dfine_criterion, dist_utils
from custom_d_fine_amd.dl import fused_optim
for mod in (fused_optim, dfine_criterion, dist_utils):
    mod.get_world_size = lambda: 2
step = bench.build_step("m", 640, dev, torch.bfloat16)
assert step.fused.overlap, "data-parallel mode not active"
ddp = timed(step)
print(f"single-process step {single:.2f} ms, data-parallel mode (1-rank RCCL group, {len(step.fused._buckets)} buckets) {ddp:.2f} ms")
dist.barrier()
dist.destroy_process_group()
