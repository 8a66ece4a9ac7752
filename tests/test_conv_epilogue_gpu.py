"""BatchNorm sums formed in the convolution store epilogue (DfineConvEpilogue, csrc/epi_bn.h): the armed convolution must
store exactly what the plain one stores, its partial sums must add up to the sums of the stored values (fp32 reference on
the same bf16 tensors), and the BatchNorm forward / backward fed with them must agree with the ones that run their own
reduction pass (tolerances: sums 2e-4 relative to the sum of magnitudes; outputs one bf16 ulp)."""
import pytest
import torch

from custom_d_fine_amd import hip

pytestmark = pytest.mark.gpu

# B, Cin, Cout, H, W: 64-channel tiles, 128-channel tiles, 256-pixel tiles over the batch (ximg), ragged channel counts,
# planes that do not fill their last tile
SHAPES = [(4, 64, 96, 40, 40), (2, 256, 512, 20, 20), (3, 128, 128, 80, 80), (2, 512, 256, 40, 40), (5, 36, 48, 24, 8),
          (2, 1024, 512, 40, 40), (1, 32, 32, 160, 160)]


def _conv_ref_sums(y):
    yf = y.float()
    return yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))


@pytest.mark.parametrize("B,Cin,Cout,H,W", SHAPES)
def test_forward_statistics_epilogue(cuda, B, Cin, Cout, H, W):
    torch.manual_seed(Cin + Cout)
    x = (torch.randn(B, Cin, H, W, device=cuda) + 0.3).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device=cuda) * Cin ** -0.5
    w2 = hip.conv_pack_weights(w, False)
    nchunk = hip.conv_epilogue_chunks(B, Cin, Cout, H, W, 1)
    assert nchunk > 0
    y0 = hip.conv_forward_bf16(x, w2, Cout, 1)
    part = hip.arm_conv_stats(Cout, nchunk, x.device)
    part.fill_(float("nan"))                                   # every slot must be written
    y1 = hip.conv_forward_bf16(x, w2, Cout, 1)
    assert torch.equal(y0, y1)
    y2 = hip.conv_forward_bf16(x, w2, Cout, 1)                 # the request is one-shot
    assert torch.equal(y0, y2)
    assert torch.isfinite(part).all()
    s, ss = _conv_ref_sums(y0)
    ps = part.double().sum(0)
    mag = y0.float().abs().sum(dim=(0, 2, 3)).double()
    assert ((ps[:, 0] - s.double()).abs() <= 2e-4 * mag + 1e-3).all()
    assert torch.allclose(ps[:, 1], ss.double(), rtol=2e-4)

    gamma, beta = torch.rand(Cout, device=cuda) + 0.5, torch.randn(Cout, device=cuda)
    for act in (None, "relu"):
        rm0, rv0 = torch.zeros(Cout, device=cuda), torch.ones(Cout, device=cuda)
        rm1, rv1 = rm0.clone(), rv0.clone()
        z0, st0 = hip.bn_act_forward(y0, gamma, beta, rm0, rv0, None, None, act, True, 0.1, 1e-5)
        z1, st1 = hip.bn_act_forward(y0, gamma, beta, rm1, rv1, None, None, act, True, 0.1, 1e-5, part=part)
        assert torch.allclose(st0, st1, rtol=2e-4, atol=2e-5)
        assert torch.allclose(rm0, rm1, rtol=1e-4, atol=1e-6) and torch.allclose(rv0, rv1, rtol=1e-4, atol=1e-6)
        d = (z0.float() - z1.float()).abs()
        assert (d <= 2 ** -7 * z0.float().abs() + 1e-3).all()


def test_forward_statistics_part_wise_conv(cuda):
    """The aggregation convolution of HG_Block reads its input as separate parts (dfine_conv1x1_seg_fwd_bf16)."""
    torch.manual_seed(5)
    B, H, W, Cout = 3, 40, 40, 64
    parts = [torch.randn(B, c, H, W, device=cuda).bfloat16() for c in (32, 16, 16, 48)]
    Cin = sum(p.shape[1] for p in parts)
    w = torch.randn(Cout, Cin, 1, 1, device=cuda) * Cin ** -0.5
    w2 = hip.conv_pack_weights(w, False)
    nchunk = hip.conv_epilogue_chunks(B, Cin, Cout, H, W, 1, len(parts))
    assert nchunk > 0
    y0 = torch.empty(B, Cout, H, W, device=cuda, dtype=torch.bfloat16)
    hip.conv1x1_seg_forward(parts, w2, (y0,))
    part = hip.arm_conv_stats(Cout, nchunk, y0.device)
    part.fill_(float("nan"))
    y1 = torch.empty_like(y0)
    hip.conv1x1_seg_forward(parts, w2, (y1,))
    assert torch.equal(y0, y1)
    s, ss = _conv_ref_sums(y0)
    ps = part.double().sum(0)
    assert torch.allclose(ps[:, 1], ss.double(), rtol=2e-4)
    assert ((ps[:, 0] - s.double()).abs() <= 2e-4 * y0.float().abs().sum(dim=(0, 2, 3)).double() + 1e-3).all()


@pytest.mark.parametrize("act,lab", [("relu", False), (None, False), ("relu", True), ("silu", True)])
@pytest.mark.parametrize("accum", [False, True])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(4, 96, 64, 40, 40), (2, 512, 256, 20, 20), (3, 128, 128, 80, 80), (5, 48, 36, 24, 8)])
def test_backward_sums_epilogue(cuda, B, Cin, Cout, H, W, act, lab, accum):
    """The consumer's data gradient (Cin -> Cout here) stores dy of a BatchNorm over Cout channels."""
    torch.manual_seed(Cin + Cout + accum)
    c = (torch.randn(B, Cout, H, W, device=cuda) * 1.5 + 0.2).bfloat16()          # the BatchNorm's input
    gamma, beta = torch.rand(Cout, device=cuda) + 0.5, torch.randn(Cout, device=cuda) * 0.3
    lab_s = torch.tensor([1.3], device=cuda) if lab else None
    lab_b = torch.tensor([0.1], device=cuda) if lab else None
    _, stats = hip.bn_act_forward(c, gamma, beta, torch.zeros(Cout, device=cuda), torch.ones(Cout, device=cuda), lab_s, lab_b,
                                  act, True, 0.1, 1e-5)
    d2 = torch.randn(B, Cin, H, W, device=cuda).bfloat16()                         # gradient arriving at the consumer
    w = torch.randn(Cin, Cout, 1, 1, device=cuda) * Cin ** -0.5                    # consumer: Cout -> Cin
    w2 = hip.conv_pack_weights(w, True)
    parked = torch.randn(B, Cout, H, W, device=cuda).bfloat16() if accum else None
    nchunk = hip.conv_epilogue_chunks(B, Cin, Cout, H, W, 1)
    assert nchunk > 0

    def dgrad(armed):
        part = None
        if armed:
            part = hip.arm_conv_bn_bwd(Cout, nchunk, c, stats, lab_s, act)
            part.fill_(float("nan"))
        if accum:
            dy = hip.conv_accumulate_bf16(d2, w2, parked.clone(), 1)
        else:
            dy = hip.conv_forward_bf16(d2, w2, Cout, 1)
        return dy, part

    dy0, _ = dgrad(False)
    dy1, part = dgrad(True)
    assert torch.equal(dy0, dy1)
    assert torch.isfinite(part).all()
    dx0, dg0, db0, dl0 = hip.bn_act_backward(c, dy0, stats, lab_s, act, True, True, lab)
    dx1, dg1, db1, dl1 = hip.bn_act_backward(c, dy0, stats, lab_s, act, True, True, lab, part=part)
    scale = dy0.float().abs().sum(dim=(0, 2, 3))
    assert ((dg0 - dg1).abs() <= 3e-4 * scale * 3 + 1e-3).all()
    assert ((db0 - db1).abs() <= 3e-4 * scale + 1e-3).all()
    if lab:
        assert torch.allclose(dl0, dl1, rtol=1e-3, atol=1e-2 * float(scale.sum()) ** 0.5)
    d = (dx0.float() - dx1.float()).abs()
    assert (d <= 2 ** -6 * dx0.float().abs() + 2e-3 * dx0.float().abs().max()).all()


@pytest.mark.parametrize("light,B,H", [(True, 16, 40), (False, 12, 80), (True, 12, 80)])
def test_hg_block_with_linked_batchnorm_sums(cuda, monkeypatch, light, B, H):
    """HG_Block end to end (ref hgnetv2.py:189-275) with DFINE_BN_LINK=1 - batch statistics from the forward convolutions'
    epilogues, backward sums from the consumers' data-gradient epilogues (kernels.BNLink) - against the default plan, which
    runs the BatchNorm's own reduction passes: same kernels otherwise, so outputs / gradients agree to bf16 rounding."""
    import os
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block

    def run(link):
        monkeypatch.setenv("DFINE_BN_LINK", link)
        kernels.reload_env()
        torch.manual_seed(3)
        blk = HG_Block(64, 32, 128, 3, residual=False, kernel_size=5 if light else 3, light_block=light, use_lab=True).to(cuda).train()
        x = torch.randn(B, 64, H, H, device=cuda).bfloat16().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x)
        go = torch.randn(y.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(5)).to(y.dtype)
        y.backward(go)
        return y.float(), x.grad.float(), [p.grad.float().clone() for p in blk.parameters()], [b.clone() for b in blk.buffers()]

    try:
        y0, gx0, gp0, bf0 = run("0")
        y1, gx1, gp1, bf1 = run("1")
    finally:
        monkeypatch.delenv("DFINE_BN_LINK", raising=False)
        kernels.reload_env()
    assert (y0 - y1).abs().max() <= 2 ** -6 * y0.abs().max()
    assert torch.nn.functional.cosine_similarity(gx0.flatten(), gx1.flatten(), dim=0) > 0.9995
    for a, b in zip(gp0, gp1):
        if a.numel() == 1:       # learnable-affine scalars: a sum of ~5 M signed terms that nearly cancels - the summation order shows
            assert (a - b).abs().max() <= 0.2 * a.abs().max() + 5e-2
        else:                    # (a 1 % change of the BatchNorm eps moves these gradients 3 - 10 x more: tools/probe/bn_link_noise.py)
            assert (a - b).abs().max() <= 5e-2 * a.abs().max() + 5e-2
    for a, b in zip(bf0, bf1):
        assert torch.allclose(a.float(), b.float(), rtol=1e-3, atol=1e-5)
