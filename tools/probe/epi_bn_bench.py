"""Cost of the BatchNorm-sum epilogues of the 1x1 convolution per layer shape (GPU box): plain / accumulate / + mode 2 sums,
forward / + mode 1 sums, next to the BatchNorm passes they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from custom_d_fine_amd import hip

dev = torch.device("cuda", 0)


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("B Cin Cout HW | conv, +accum, +accum+sums2, conv+sums2 | bn_bwd, bn_bwd(part) || conv+sums1 | bn_fwd, bn_fwd(part)   [us]")
for B, Cin, Cout, H in [(32, 128, 128, 40), (32, 128, 384, 40), (32, 384, 768, 40), (32, 64, 96, 80), (32, 192, 384, 80), (32, 48, 96, 160),
                        (32, 128, 512, 40), (32, 256, 256, 80)]:
    W = H
    # data gradient of a consumer Cout <- Cin ... here: conv Cin -> Cout whose output is dy of a BatchNorm over Cout channels
    x = torch.randn(B, Cin, H, W, device=dev).bfloat16()
    w2 = hip.conv_pack_weights(torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5, False)
    c = torch.randn(B, Cout, H, W, device=dev).bfloat16()
    g, bt = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    _, stats = hip.bn_act_forward(c, g, bt, rm, rv, None, None, "relu", True, 0.1, 1e-5)
    y = torch.empty(B, Cout, H, W, device=dev, dtype=torch.bfloat16)
    nch = hip.conv_epilogue_chunks(B, Cin, Cout, H, W, 1)
    t0 = t(lambda: hip.conv_forward_bf16(x, w2, Cout, 1))
    t1 = t(lambda: hip.conv_accumulate_bf16(x, w2, y, 1))

    def acc2():
        hip.arm_conv_bn_bwd(Cout, nch, c, stats, None, "relu")
        hip.conv_accumulate_bf16(x, w2, y, 1)

    def plain2():
        hip.arm_conv_bn_bwd(Cout, nch, c, stats, None, "relu")
        hip.conv_forward_bf16(x, w2, Cout, 1)

    def fwd1():
        hip.arm_conv_stats(Cout, nch, dev)
        hip.conv_forward_bf16(x, w2, Cout, 1)

    t2, t3, t4 = t(acc2), t(plain2), t(fwd1)
    part2 = hip.arm_conv_bn_bwd(Cout, nch, c, stats, None, "relu")
    dy = hip.conv_forward_bf16(x, w2, Cout, 1)
    part1 = hip.arm_conv_stats(Cout, nch, dev)
    hip.conv_forward_bf16(x, w2, Cout, 1)
    tb0 = t(lambda: hip.bn_act_backward(c, dy, stats, None, "relu", True, True, False))
    tb1 = t(lambda: hip.bn_act_backward(c, dy, stats, None, "relu", True, True, False, part=part2))
    tf0 = t(lambda: hip.bn_act_forward(c, g, bt, rm, rv, None, None, "relu", True, 0.1, 1e-5))
    tf1 = t(lambda: hip.bn_act_forward(c, g, bt, rm, rv, None, None, "relu", True, 0.1, 1e-5, part=part1))
    print(f"{B} {Cin:4d} {Cout:4d} {H:3d} | {t0:6.1f} {t1:6.1f} {t2:6.1f} {t3:6.1f} | {tb0:6.1f} {tb1:6.1f} || {t4:6.1f} | {tf0:6.1f} {tf1:6.1f}   nchunk {nch}")
