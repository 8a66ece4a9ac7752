"""A1/A2 parity of the row-streaming 3x3 weight-gradient kernel (csrc/wgrad3.hip, through dfine_conv_wgrad_bf16) against the
fp32 convolution weight gradient of torch on the SAME bf16-rounded operands (reference call site: the autograd of
nn.Conv2d(k=3, s=1, p=1) in /root/reference/src/d_fine/arch/hgnetv2.py:35-80 and hybrid_encoder.py:21-156).
Tolerance: fp32 accumulation in another order - 2e-3 of the largest gradient element."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W   - every tile configuration, channel / row remainders, one-row and one-unit images
    (2, 128, 128, 80, 80), (3, 128, 128, 40, 40), (2, 128, 128, 20, 24), (2, 64, 64, 80, 80), (3, 96, 64, 80, 80),
    (2, 32, 32, 160, 160), (2, 32, 16, 9, 160), (1, 64, 64, 5, 8), (2, 256, 256, 40, 40), (2, 128, 128, 7, 120),
    (1, 48, 80, 1, 16), (2, 64, 128, 3, 40), (33, 128, 128, 10, 16), (2, 16, 32, 31, 104), (1, 160, 144, 23, 56),
]


def _reference(x, dy):
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, device=x.device, requires_grad=True)
    F.conv2d(x.float(), w, padding=1).backward(dy.float())
    return w.grad


@pytest.mark.parametrize("B,Cin,Cout,H,W", CASES)
def test_wgrad3_matches_fp32_reference(cuda, B, Cin, Cout, H, W):
    from custom_d_fine_amd import hip
    torch.manual_seed(B * 1000 + Cin + H)
    x = torch.randn(B, Cin, H, W, device=cuda).bfloat16()
    dy = torch.randn(B, Cout, H, W, device=cuda).bfloat16()
    ref = _reference(x, dy)
    got = hip.conv_wgrad_bf16(x, dy, 3)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == torch.float32
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-3 * scale, ((got - ref).abs().max().item(), scale)


def test_wgrad3_partials_layout(cuda):
    """The deferred path: [splits][NP16][CP16][9] partial sums whose sum over the splits is the gradient (what
    dfine_multi_wgrad_reduce adds into the flat gradient buffer)."""
    from custom_d_fine_amd import hip
    torch.manual_seed(3)
    x = torch.randn(4, 96, 40, 40, device=cuda).bfloat16()
    dy = torch.randn(4, 64, 40, 40, device=cuda).bfloat16()
    ws, (splits, cout, cin, taps, np16, cp16) = hip.conv_wgrad_bf16(x, dy, 3, partials=True)
    hip.side_join()
    torch.cuda.synchronize()
    assert (cout, cin, taps, np16, cp16) == (64, 96, 9, 64, 96) and ws.numel() == splits * np16 * cp16 * 9
    got = ws.view(splits, np16, cp16, 3, 3).sum(0)[:cout, :cin]
    ref = _reference(x, dy)
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_wgrad3_one_hot_taps(cuda):
    """Transpose / shift detector: one hot pixel in x and one in dy -> exactly one tap of one (n, c) pair, for every tap and for
    pixels on the map's border (where a wrong pad would leak the neighbouring row)."""
    from custom_d_fine_amd import hip
    H, W = 6, 16
    for (xr, xc) in ((0, 0), (2, 7), (5, 15), (3, 8), (0, 15)):
        for kr in range(3):
            for kc in range(3):
                yr, yc = xr - kr + 1, xc - kc + 1                  # output pixel that sees x[xr, xc] under tap (kr, kc)
                if not (0 <= yr < H and 0 <= yc < W):
                    continue
                x = torch.zeros(1, 40, H, W, device=cuda)
                dy = torch.zeros(1, 70, H, W, device=cuda)
                x[0, 33, xr, xc] = 2.0
                dy[0, 65, yr, yc] = 3.0
                got = hip.conv_wgrad_bf16(x.bfloat16(), dy.bfloat16(), 3)
                want = torch.zeros_like(got)
                want[65, 33, kr, kc] = 6.0
                assert torch.equal(got, want), (xr, xc, kr, kc, got.nonzero().tolist())
