"""One steady-state D-FINE-m bs=32 train step under torch.profiler: device time per kernel family (GPU box)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch

FAMILIES = (("gn_", "mask: groupnorm"), ("bilinear_", "mask: bilinear"), ("mask_", "mask: losses / costs"), ("conv1x1", "conv1x1 fwd/dgrad"), ("conv_wgrad1_glds", "conv wgrad 1x1"), ("conv_wgrad1_group", "conv wgrad 1x1"), ("conv_wgrad_kernel<1>", "conv wgrad 1x1"),
            ("conv_wgrad_kernel<3>", "conv wgrad 3x3"), ("conv_wgrad_reduce", "wgrad reduce"), ("conv_igemm", "conv3x3 fwd/dgrad"), ("conv3x3_ws", "conv3x3 fwd/dgrad"), ("conv3x3_rows32", "conv3x3 fwd/dgrad"), ("conv_wgrad3_rows", "conv wgrad 3x3"), ("multi_wgrad_reduce", "wgrad reduce"), ("fx_", "msda"), ("cast_f16acc", "msda"),
            ("dfine::bn_", "bn"), ("msda", "msda"), ("cast_f32_bf16", "msda"), ("linear_act", "linear_act"), ("linear_ring", "linear_act"), ("act_", "linear_act"),
            ("linear_wgrad", "linear_wgrad"), ("attn_", "attention"), ("stem_", "stem"), ("stem3_", "stem"), ("dwconv", "dwconv"), ("ln_fused", "ln"),
            ("dfine::", "HIP other"), ("Cijk", "hipBLASLt"), ("igemm", "MIOpen"), ("batched_transpose", "MIOpen"), ("CatArray", "ATen cat"),
            ("FillFunctor", "ATen fill"), ("copy", "ATen copy/cast"), ("Memcpy", "ATen copy/cast"), ("CUDAFunctor_add", "ATen add"),
            ("at::native", "ATen other"))
dev = torch.device("cuda", 0)
MODEL, IMG, BATCH, MASK = os.environ.get("SP_MODEL", "m"), int(os.environ.get("SP_IMG", "640")), int(os.environ.get("SP_BATCH", "32")), os.environ.get("SP_MASK", "0") == "1"
step = bench.build_step(MODEL, IMG, dev, torch.bfloat16, mask=MASK)
images, targets = make_batch(BATCH, IMG, seed=42, device=dev, with_masks=MASK)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step(images, targets)
    torch.cuda.synchronize()
agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
for k in prof.key_averages():
    fam = next((f for pat, f in FAMILIES if pat in k.key), "other")
    agg[fam] += k.device_time_total / 3e3
    cnt[fam] += k.count // 3
tot = sum(agg.values())
print(f"D-FINE-{MODEL}{'+mask' if MASK else ''} {IMG}x{IMG} bs {BATCH}: device time per step {tot:.2f} ms, {sum(cnt.values())} launches  (env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DFINE_")) + ")")
for f, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {f:22s} {v:7.2f} ms  {cnt[f]:5d} launches")
if os.environ.get("STEP_PROFILE_TOP"):
    for k in sorted(prof.key_averages(), key=lambda k: -k.device_time_total)[:int(os.environ["STEP_PROFILE_TOP"])]:
        print(f"{k.device_time_total/3e3:7.2f} {k.count//3:5d} {k.key[:120]}")
if os.environ.get("STEP_PROFILE_OTHER"):
    for k in sorted(prof.key_averages(), key=lambda k: -k.count):
        fam = next((f for pat, f in FAMILIES if pat in k.key), "other")
        if fam in ("other", "ATen copy/cast", "ATen other", "ATen fill") and k.count >= 30:
            print(f"{k.count//3:5d} x {k.device_time_total/3e3:7.3f} ms  {k.key[:150]}")
