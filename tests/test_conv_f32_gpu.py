"""fp32 dense convolutions on the f32-input MFMA kernels (csrc/conv_f32.hip, BASELINE configs[1]) through the C ABI against
F.conv2d in fp64 -> fp32 (forward, data gradient, weight gradient): exact fp32 arithmetic, so the tolerance is accumulation
order only (1e-5 of the output scale)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from custom_d_fine_amd import kernels

pytestmark = pytest.mark.gpu

CASES = [  # cin, cout, k, stride, pad, H, W, pad_br
    (3, 16, 3, 2, 1, 64, 96, False),        # stem1: 3 input channels, stride 2
    (16, 8, 2, 1, 0, 33, 47, True),         # stem2a: 2x2 on the bottom / right padded map
    (24, 32, 3, 2, 1, 40, 40, False),       # stem3-like, stride 2 with a data gradient
    (32, 48, 1, 1, 0, 40, 40, False),
    (64, 64, 3, 1, 1, 20, 20, False),
    (40, 72, 3, 1, 1, 17, 23, False),       # ragged channels / odd sizes
    (128, 128, 1, 1, 0, 16, 320, False),    # wider than one 160-pixel column tile
    (16, 16, 3, 1, 1, 8, 200, False),       # ragged column tiles
]


@pytest.mark.parametrize("cin,cout,k,s,p,H,W,pad_br", CASES)
def test_conv_f32_matches_conv2d(cuda, cin, cout, k, s, p, H, W, pad_br):
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, s, p, bias=False).to(cuda)
    x = torch.randn(2, cin, H, W, device=cuda, requires_grad=True)
    assert kernels._f32_conv_ok(conv, x)
    y = kernels.conv_f32(x, conv, pad_br)
    go = torch.randn_like(y)
    y.backward(go)
    got = (y.detach(), x.grad.clone(), conv.weight.grad.clone())
    xr = x.detach().double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)) if pad_br else xr, wr, None, s, p)
    assert yr.shape == y.shape
    yr.backward(go.double())
    for a, b, name in ((got[0], yr.detach(), "y"), (got[1], xr.grad, "dx"), (got[2], wr.grad, "dw")):
        err = (a.double() - b).abs().max().item()
        assert err <= 2e-5 * max(b.abs().max().item(), 1e-6), (name, err, b.abs().max().item())


def test_conv_bn_act_takes_the_f32_kernel_without_autocast(cuda, monkeypatch):
    from custom_d_fine_amd import hip
    calls = []
    real = hip.conv_f32_forward
    monkeypatch.setattr(hip, "conv_f32_forward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    conv, bn = nn.Conv2d(32, 32, 3, 1, 1, bias=False).to(cuda), nn.BatchNorm2d(32).to(cuda)
    x = torch.randn(2, 32, 24, 24, device=cuda)
    y = kernels.conv_bn_act(x, conv, bn, "relu", None)
    want = F.relu(bn(conv(x)))
    assert len(calls) == 1 and y.dtype == torch.float32
    assert (y - want).abs().max() <= 1e-4 * want.abs().max()


def test_conv1x1_f32_reads_channel_slices_in_place(cuda, monkeypatch):
    """A 1x1 convolution on a channel slice of a concatenation, with a gradient that is a slice of the concatenation's gradient:
    forward, data gradient and weight gradient read the slices in place (no contiguous copies) and match conv2d."""
    from custom_d_fine_amd import hip
    torch.manual_seed(5)
    conv = nn.Conv2d(32, 48, 1, bias=False).to(cuda)
    full = torch.randn(3, 56, 20, 20, device=cuda, requires_grad=True)
    x = full[:, 8:40]
    assert not x.is_contiguous() and hip.planes_dense(x)
    y = kernels.conv_f32(x, conv)
    gfull = torch.randn(3, 64, 20, 20, device=cuda)
    go = gfull[:, 4:52]
    assert hip.planes_dense(go) and not go.is_contiguous()
    seen = []
    real = hip.conv1x1_f32
    monkeypatch.setattr(hip, "conv1x1_f32", lambda a, w: (seen.append(a.is_contiguous()), real(a, w))[1])
    y.backward(go)
    assert seen == [False]                                                   # the data gradient took the slice itself
    xr = full.detach().double()[:, 8:40].requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(go.double())
    for a, b, name in ((y.detach(), yr.detach(), "y"), (full.grad[:, 8:40], xr.grad, "dx"), (conv.weight.grad, wr.grad, "dw")):
        err = (a.double() - b).abs().max().item()
        assert err <= 2e-5 * max(b.abs().max().item(), 1e-6), (name, err)
    assert full.grad[:, :8].abs().max().item() == 0.0 and full.grad[:, 40:].abs().max().item() == 0.0
