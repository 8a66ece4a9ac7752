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
    # weight gradient (MFMA split-K kernel, or MIOpen when W % 8 != 0)
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
