"""d(value) accumulation of the deformable-attention backward: f32 atomics (0) vs scaled f16 with one packed atomic per channel pair (2) vs int32 fixed point with channel pairs in one 64-bit integer atomic (3)
at the bench shape (D-FINE-m, 640x640, bs 32, Lq = 492 = 300 selected + 192 denoising queries clustered around the targets),
interleaved rounds in one process; also prints each mode's error against an fp64 accumulation of the same contributions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_d_fine_amd import hip

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, Lq, H, D, L = 32, 492, 8, 32, 8400
shapes, points = ((80, 80), (40, 40), (20, 20)), (3, 6, 3)
value = torch.randn(B, L, H, D, device=dev, dtype=torch.bfloat16)
gt = torch.cat([torch.rand(B, 7, 2, device=dev) * 0.6 + 0.2, torch.rand(B, 7, 2, device=dev) * 0.3 + 0.05], -1)
dn = gt.repeat(1, 28, 1)[:, :192] + torch.randn(B, 192, 4, device=dev) * 0.02          # noised copies of the targets
if os.environ.get("MSDA_PADS", "1") == "1":
    # a training batch pads every denoising group to the largest target count of the batch: with 7 targets of at most 16 the
    # last 9 entries of every 16 are zero boxes (reference points sigmoid(inverse_sigmoid(0)) = 1e-5: all points on pixel (0, 0))
    dn = dn.view(B, 12, 16, 4).clone()
    dn[:, :, 7:] = 1e-5
    dn = dn.view(B, 192, 4)
ref = torch.cat([dn.clamp(1e-5, 0.99), torch.cat([torch.rand(B, 300, 2, device=dev), torch.rand(B, 300, 2, device=dev) * 0.3 + 0.02], -1)], 1).contiguous()
off = (torch.randn(B, Lq, H, 12, 2, device=dev) * 2).bfloat16()
lg = torch.randn(B, Lq, H, 12, device=dev).bfloat16()
go = torch.randn(B, Lq, H * D, device=dev, dtype=torch.bfloat16)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
times = {0: [], 2: [], 3: []}
outs = {}
for r in range(rounds + 2):
    for mode in (0, 2, 3):
        hip.MSDA_ACC_MODE = mode
        acc = hip.msda_grad_value_buffer(value)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        hip.msda_fused_backward(value, ref, off, lg, go, shapes, points, 0.5, gv_acc=acc)
        b.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[mode].append(a.elapsed_time(b) * 1e3)
        if r == 0:
            outs[mode] = hip.msda_finish_grad_value(acc, torch.float32).float()
for mode in (0, 2, 3):
    t = sorted(times[mode])
    print(f"mode {mode}: median {t[len(t) // 2]:.1f} us, min {t[0]:.1f} us")
ref32 = outs[0]
for mode in (2, 3):
    d = (outs[mode] - ref32).abs()
    print(f"mode {mode} vs f32 atomics: max abs {d.max().item():.4g} (|ref| max {ref32.abs().max().item():.4g}), "
          f"rel-to-max {d.max().item() / ref32.abs().max().item():.3g}, mean rel {(d.sum() / ref32.abs().sum()).item():.3g}")
hits = (ref32 != 0).float().mean().item()
print(f"non-zero fraction of d(value): {hits:.3f}")
