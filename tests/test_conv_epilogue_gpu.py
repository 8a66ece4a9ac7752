"""A1: accumulate epilogue of the dense convolutions - a data gradient added onto an existing one (gradient fan-in of a
map with two consumers inside HG_Block, ref hgnetv2.py:265-274) - against the plain kernel + a separate bf16 add
(bit-exact), and the HG_Block built on it against autograd's own gradient sums."""
import pytest
import torch

from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block

pytestmark = pytest.mark.gpu

SHAPES = [  # B, Cin, Cout, H, W, KS
    (4, 128, 128, 40, 40, 1), (3, 64, 256, 20, 20, 1), (2, 256, 384, 80, 80, 1), (2, 96, 40, 20, 20, 1),
    (2, 768, 1536, 20, 20, 1), (2, 64, 64, 80, 80, 3), (3, 128, 128, 20, 20, 3), (2, 32, 32, 160, 160, 3),
    (2, 128, 128, 40, 40, 3), (5, 96, 64, 80, 80, 3)]


def _hip():
    from custom_d_fine_amd import hip
    return hip


def _case(cuda, B, cin, cout, H, W, ks, seed=0):
    g = torch.Generator().manual_seed(seed + ks)
    x = (torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(cuda)
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (cin * ks * ks) ** -0.5).to(cuda)
    return x, w


@pytest.mark.parametrize("B,cin,cout,H,W,ks", SHAPES)
def test_conv_accumulate_equals_separate_add(cuda, B, cin, cout, H, W, ks):
    hip = _hip()
    if not hip.conv_epilogue_supported(B, cin, cout, H, W, ks):
        pytest.skip("shape runs on the kernels without the epilogue")
    x, w = _case(cuda, B, cin, cout, H, W, ks, seed=3)
    w2 = hip.conv_pack_weights(w.float().contiguous(), False)
    base = (torch.randn(B, cout, H, W) * 2).to(torch.bfloat16).to(cuda)
    want = base + hip.conv_forward_bf16(x, w2, cout, ks)             # bf16 + bf16 -> bf16 (one rounding), like autograd's sum
    got = hip.conv_accumulate_bf16(x, w2, base.clone(), ks)
    assert torch.equal(got, want)


def test_hg_block_gradient_fanin_matches_autograd_sum(cuda, monkeypatch):
    """HG_Block (3x3 units and light units): data gradients accumulated in the convolution epilogues (DFINE_GRAD_FANIN=1)
    against autograd's own sums (=0): same kernels and the same bf16 additions: the input gradient agrees bit for bit."""
    for light, cin, mid, cout, hw, k in [(False, 64, 32, 128, 80, 3), (True, 128, 64, 256, 40, 5)]:
        torch.manual_seed(5)
        blk = HG_Block(cin, mid, cout, layer_num=3, kernel_size=k, residual=False, light_block=light, use_lab=True).to(cuda).train()
        x0 = torch.randn(2, cin, hw, hw).to(cuda)
        go = torch.randn(2, cout, hw, hw).to(cuda)
        res = []
        for flag in ("0", "1"):
            monkeypatch.setenv("DFINE_GRAD_FANIN", flag)
            kernels.reload_env()
            blk.zero_grad()
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x)
            y.backward(go.to(y.dtype))
            torch.cuda.synchronize()
            res.append([x.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None])
        monkeypatch.delenv("DFINE_GRAD_FANIN")
        kernels.reload_env()
        assert len(res[0]) == len(res[1])
        assert torch.equal(res[0][0], res[1][0])                     # the input gradient: same kernels, same bf16 additions
        for a, b in zip(res[0][1:], res[1][1:]):                     # (parameter gradients: reductions with atomics in the BN tail)
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()))


@pytest.mark.parametrize("shape", [(2, 16, 20, 24), (3, 256, 40, 40), (1, 8, 5, 8), (2, 256, 20, 20), (1, 4, 3, 12)])
def test_nearest_upsample_2x_matches_interpolate(cuda, shape):
    """FPN top-down upsampling (ref hybrid_encoder.py:472): forward bit-exact (pure copy), backward = fp32 sum of the 2 x 2
    block rounded once to bf16 (ATen's bf16 backward accumulates in fp32 as well)."""
    import torch.nn.functional as F
    torch.manual_seed(1)
    x = torch.randn(shape, device=cuda).bfloat16().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = kernels.upsample2_nearest(x)
    yr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    assert torch.equal(y, yr)
    go = torch.randn_like(yr)
    y.backward(go)
    yr.backward(go)
    want = go.float().view(shape[0], shape[1], shape[2], 2, shape[3], 2).sum((3, 5)).bfloat16()
    assert torch.equal(x.grad, want)
    assert (x.grad.float() - xr.grad.float()).abs().max() <= 2 ** -7 * xr.grad.float().abs().max()


def test_multi_pack_equals_single_pack(cuda):
    """dfine_conv_pack_weights_multi (all layers in one launch; the data-gradient packing as 32 x 32 tile transposes through LDS)
    against the element-wise single-layer kernel, bit for bit: 1x1 and 3x3, forward and data-gradient forms, channel counts
    that are not multiples of the tile."""
    from custom_d_fine_amd import hip
    torch.manual_seed(0)
    shapes = [(512, 384, 1), (128, 96, 3), (43, 21, 3), (80, 256, 1), (24, 48, 3), (1536, 768, 1)]
    rows, outs, keep = [], [], []
    for cout, cin, ks in shapes:
        w = torch.randn(cout, cin, ks, ks, device=cuda)
        keep.append(w)
        for dgrad in (False, True):
            co, ci = (cin, cout) if dgrad else (cout, cin)
            n = hip.conv_packed_elems(cout, cin, ks, dgrad)
            dst = torch.full((n,), -1, device=cuda, dtype=torch.bfloat16)
            NP, KP = (co + 15) // 16 * 16, (ci + 31) // 32 * 32
            assert n == ks * ks * NP * KP
            rows.append([w.data_ptr(), dst.data_ptr(), cout, cin, ks, NP, KP, int(dgrad)])
            outs.append((dst, hip.conv_pack_weights(w, dgrad)))
    table = torch.tensor(rows, dtype=torch.int64, device=cuda)
    hip.conv_pack_weights_multi(table, len(rows))
    torch.cuda.synchronize()
    for got, want in outs:
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
