"""A1/A2 parity: implicit-GEMM MFMA convolution (forward + data gradient through the C ABI) vs a
plain PyTorch fp32 reference of the same op on the bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from custom_d_fine_amd import kernels

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W, KS
    (2, 32, 32, 16, 160, 3), (2, 128, 128, 40, 40, 3), (3, 96, 64, 80, 80, 3), (2, 128, 128, 20, 20, 3),
    (2, 64, 48, 13, 22, 3), (2, 128, 128, 40, 40, 1), (2, 160, 48, 160, 160, 1), (2, 352, 192, 80, 80, 1),
    (3, 48, 96, 20, 20, 1), (1, 1792, 768, 20, 20, 1), (2, 256, 128, 30, 30, 1), (2, 128, 256, 7, 10, 1),
    (2, 512, 512, 80, 80, 1), (2, 768, 256, 40, 40, 1), (1, 1536, 256, 20, 20, 1), (2, 96, 48, 8, 8, 1), (2, 160, 80, 24, 24, 1),
    (33, 128, 128, 8, 16, 1), (5, 768, 1536, 20, 20, 1), (3, 1280, 384, 40, 40, 1), (7, 256, 256, 20, 20, 1), (9, 64, 96, 12, 10, 1),
    # <= 32 channels on both sides, W % 16 == 0: the row-streaming kernel of csrc/conv3s.hip (rows per workgroup, remainders, one row)
    (2, 32, 32, 160, 160, 3), (3, 16, 32, 9, 48, 3), (2, 32, 16, 33, 64, 3), (1, 16, 16, 1, 32, 3), (5, 32, 32, 17, 144, 3),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,KS", CASES)
def test_conv_fwd_and_dgrad(cuda, B, Cin, Cout, H, W, KS):
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, device=cuda).bfloat16()
    w = (torch.randn(Cout, Cin, KS, KS, device=cuda) / (Cin * KS * KS) ** 0.5)
    wb = w.bfloat16().float()
    xr = x.float().requires_grad_(True)
    yr = F.conv2d(xr, wb, padding=KS // 2)
    go = torch.randn_like(yr).bfloat16()
    yr.backward(go.float())
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    y = kernels._DenseConvMFMA.apply(xg, wg)
    assert y.dtype == torch.bfloat16 and y.shape == yr.shape
    y.backward(go)
    scale = yr.abs().max().item()
    assert (y.float() - yr).abs().max().item() < 1.5e-2 * scale
    gscale = xr.grad.abs().max().item()
    assert (xg.grad.float() - xr.grad).abs().max().item() < 1.5e-2 * gscale
    # weight gradient (MFMA split-K kernel; zero-padded copies when W % 8 != 0 (3x3) / H * W % 8 != 0 (1x1))
    wr = wb.clone().requires_grad_(True)
    F.conv2d(x.float(), wr, padding=KS // 2).backward(go.float())
    wscale = wr.grad.abs().max().item()
    assert (wg.grad - wr.grad).abs().max().item() < 1.5e-2 * wscale


def test_asymmetric_layout_check(cuda):
    """Transpose-detecting check: one hot input channel / pixel and asymmetric weights."""
    x = torch.zeros(1, 32, 8, 16, device=cuda)
    x[0, 5, 3, 7] = 1.0
    w = torch.arange(48 * 32 * 9, device=cuda, dtype=torch.float32).reshape(48, 32, 3, 3) / 1000.0
    y = kernels._DenseConvMFMA.apply(x.bfloat16(), w)
    ref = F.conv2d(x, w.bfloat16().float(), padding=1)
    assert torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2)


def test_packed_weight_caches_follow_weight_updates(cuda):
    """The packed-weight / bf16-shadow caches must notice (a) in-place torch updates (tensor._version) and (b) raw-pointer
    updates by the fused optimizer kernels, which only `bump_weight_epoch()` announces."""
    from custom_d_fine_amd import hip, kernels
    torch.manual_seed(0)
    x = torch.randn(2, 64, 16, 16, device=cuda).to(torch.bfloat16)
    w = torch.nn.Parameter(torch.randn(64, 64, 1, 1, device=cuda) * 0.1)
    lin_w = torch.nn.Parameter(torch.randn(32, 64, device=cuda) * 0.1)

    def run():
        with torch.no_grad():
            return kernels._DenseConv.apply(x, w).float()

    kernels._CONV_PLAN.clear()
    import os
    kernels.reload_env()
    try:
        y1 = run()
        y1b = run()                                           # served from the cache
        assert torch.equal(y1, y1b)
        with torch.no_grad():
            w.mul_(2.0)                                       # (a) in-place: _version changes
        y2 = run()
        assert torch.allclose(y2, 2 * y1, rtol=2e-2, atol=1e-3)
        # (b) raw-pointer update (what adamw_ema_kernel does): ema := 0 * ema + 1 * src through the C ABI
        src = (w.detach() * 0.5).contiguous()
        hip.ema_update(w.detach().view(-1), src.view(-1), 0.0)
        stale = run()
        assert torch.allclose(stale, y2)                      # nothing announced the change yet
        kernels.bump_weight_epoch()
        y3 = run()
        assert torch.allclose(y3, y1, rtol=2e-2, atol=1e-3)
        # bf16 shadows follow the same protocol
        b1 = kernels.bf16_param(lin_w).clone()
        with torch.no_grad():
            lin_w.add_(1.0)
        assert torch.allclose(kernels.bf16_param(lin_w).float(), b1.float() + 1.0, atol=2e-2)
    finally:
        kernels.reload_env()


@pytest.mark.parametrize("chans,Cout,H,W", [((128, 64, 64, 64), 256, 40, 40), ((96, 32, 32), 128, 80, 80), ((256, 256), 256, 20, 20),
                                            ((48, 16, 16, 16, 16, 16, 16), 96, 16, 24)])
def test_segmented_conv1x1_matches_cat_conv(cuda, chans, Cout, H, W):
    """1x1 conv over torch.cat(parts) with the parts read in place (ChanSegs): forward, per-part data gradients and the
    weight gradient vs fp32 torch on the concatenation; one part is a channel slice of a wider tensor."""
    torch.manual_seed(sum(chans) + Cout)
    B = 3
    wide = torch.randn(B, chans[0] + 32, H, W, device=cuda).bfloat16()
    parts = [wide[:, 32:]] + [torch.randn(B, c, H, W, device=cuda).bfloat16() for c in chans[1:]]
    assert not parts[0].is_contiguous()
    parts = [p.requires_grad_(True) for p in parts]
    w = (torch.randn(Cout, sum(chans), 1, 1, device=cuda) / sum(chans) ** 0.5).requires_grad_(True)
    y = kernels._DenseConvSeg.apply(w, None, *parts)
    go = torch.randn_like(y)
    y.backward(go)
    pr = [p.detach().float().requires_grad_(True) for p in parts]
    wr = w.detach().bfloat16().float().requires_grad_(True)
    yr = F.conv2d(torch.cat(pr, 1), wr)
    yr.backward(go.float())
    assert (y.float() - yr).abs().max().item() < 1.5e-2 * yr.abs().max().item()
    for p, r in zip(parts, pr):
        assert p.grad.shape == r.grad.shape
        assert (p.grad.float() - r.grad).abs().max().item() < 1.5e-2 * r.grad.abs().max().item()
    assert (w.grad - wr.grad).abs().max().item() < 1.5e-2 * wr.grad.abs().max().item()
