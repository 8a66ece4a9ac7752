"""HGNetv2 stem kernels (csrc/stem.hip) against the ATen composition of the reference's StemBlock
(src/d_fine/arch/hgnetv2.py:115-166): F.pad + conv2d + max_pool2d in fp32 on the same bf16 inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


# (Cin, Cout, ks, stride, pad, pad_br): the five layers of the B2 (D-FINE-m) and B0 (n / s) stems
LAYERS = [(3, 24, 3, 2, 1, False), (24, 12, 2, 1, 0, True), (12, 24, 2, 1, 0, True), (48, 24, 3, 2, 1, False),
          (24, 32, 1, 1, 0, False), (3, 16, 3, 2, 1, False), (16, 8, 2, 1, 0, True), (8, 16, 2, 1, 0, True),
          (32, 16, 3, 2, 1, False), (16, 16, 1, 1, 0, False)]


@pytest.mark.parametrize("cin,cout,ks,stride,pad,pad_br", LAYERS)
@pytest.mark.parametrize("hw", [(64, 128), (40, 64)])
def test_stem_conv_forward_backward(cuda, cin, cout, ks, stride, pad, pad_br, hw):
    from custom_d_fine_amd import kernels
    torch.manual_seed(cin * 100 + cout)
    H, W = hw
    need_dx = cin != 3                    # stem1 reads the image: no data gradient exists for it
    x = torch.randn(3, cin, H, W, device=cuda).to(torch.bfloat16).requires_grad_(need_dx)
    w = (torch.randn(cout, cin, ks, ks, device=cuda) / (cin * ks * ks) ** 0.5).requires_grad_(True)
    y = kernels._StemConv.apply(x, w, stride, pad, pad_br)
    xr = x.detach().float().requires_grad_(need_dx)
    wr = w.detach().clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)) if pad_br else xr, wr, None, stride, pad)
    assert y.shape == yr.shape and y.dtype == torch.bfloat16
    assert _rel(y, yr) < 1e-2                                   # bf16 rounding of the output
    g = torch.randn_like(yr).to(torch.bfloat16)
    y.backward(g)
    yr.backward(g.float())
    if need_dx:
        assert _rel(x.grad, xr.grad) < 1e-2
    assert _rel(w.grad, wr.grad) < 2e-3                         # fp32 accumulation of bf16 products


def test_stem_conv_without_input_grad(cuda):
    """stem1 sees the image: no data gradient is requested (and none exists for 3 -> 24 stride 2)."""
    from custom_d_fine_amd import kernels
    x = torch.randn(2, 3, 64, 64, device=cuda).to(torch.bfloat16)
    w = torch.randn(24, 3, 3, 3, device=cuda, requires_grad=True)
    y = kernels._StemConv.apply(x, w, 2, 1, False)
    y.float().square().sum().backward()
    wr = w.detach().clone().requires_grad_(True)
    F.conv2d(x.float(), wr, None, 2, 1).to(torch.bfloat16).float().square().sum().backward()
    assert _rel(w.grad, wr.grad) < 1e-2


@pytest.mark.parametrize("shape", [(2, 24, 64, 96), (1, 5, 33, 47)])
def test_stem_pool_matches_aten_bit_exact(cuda, shape):
    from custom_d_fine_amd import kernels
    torch.manual_seed(3)
    # post-ReLU-like data with a learnable-affine offset: many exact ties (zeros -> equal values), some negative
    x = (torch.relu(torch.randn(shape, device=cuda)) * 0.7 - 0.1).to(torch.bfloat16).requires_grad_(True)
    y = kernels._StemPool.apply(x)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.max_pool2d(F.pad(xr, (0, 1, 0, 1)), kernel_size=2, stride=1, ceil_mode=True)
    assert torch.equal(y, yr)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    # ATen accumulates the (up to four) window gradients of a pixel in fp32 atomics and rounds once, like here
    assert (x.grad.float() - xr.grad.float()).abs().max().item() <= 2 ** -6 * g.abs().max().item()
    assert ((x.grad != 0) == (xr.grad != 0)).all()              # same argmax choice on ties


def test_stem_pool_backward_onto_parked_gradient(cuda):
    """StemBlock's stem1 output feeds the pool and stem2a (ref hgnetv2.py:158-163): the pool's backward adds its gradient onto
    stem2a's parked one - bit-identical to autograd's separate add of the two bf16 maps."""
    from custom_d_fine_amd import hip, kernels
    torch.manual_seed(4)
    x = (torch.relu(torch.randn(3, 24, 40, 64, device=cuda)) * 0.7 - 0.1).to(torch.bfloat16)
    dy = torch.randn_like(x)
    other = torch.randn_like(x)
    want = hip.stem_pool_backward(x, dy) + other
    got = other.clone()
    assert hip.stem_pool_backward(x, dy, acc=got) is got
    assert torch.equal(got, want)

    # through autograd, eager: pool created first -> its backward runs after the other consumer's
    xa = x.clone().requires_grad_(True)
    fan = kernels.GradFanIn()
    with torch.autocast("cuda", dtype=torch.bfloat16):         # the stem kernels serve the autocast path
        p = kernels.stem_pool(xa, fanin=fan)
        q = kernels.park_grad(xa, fan, owned=True) * 1.5
        assert fan.armed and fan.parking
        torch.autograd.backward([p, q], [dy, other])
        xb = x.clone().requires_grad_(True)
        torch.autograd.backward([kernels.stem_pool(xb), xb * 1.5], [dy, other])
    assert torch.equal(xa.grad, xb.grad)


def test_stem_block_matches_aten_composition(cuda):
    """Whole StemBlock under bf16 autocast: the HIP path and the ATen bf16 path of the same module
    (DFINE_STEM=0) are compared with the module run in fp32; bf16 rounding flips ReLU / max-pool choices,
    so the two bf16 runs are judged by their distance to the fp32 result, not to each other."""
    import os
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import StemBlock
    torch.manual_seed(0)
    blk = StemBlock(3, 24, 32, use_lab=True).to(cuda).train()
    x = torch.randn(2, 3, 128, 128, device=cuda)

    def run(flag, amp):
        os.environ["DFINE_STEM"] = flag
        os.environ["DFINE_ALLOW_LIBRARY"] = "1" if flag == "0" else "0"      # the ATen / MIOpen composition is the comparison run
        kernels.reload_env()
        blk.zero_grad()
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = blk(x)
        y.float().square().mean().backward()
        return y.detach().float(), {n: p.grad.detach().float().clone() for n, p in blk.named_parameters()}

    try:
        y_hip, g_hip = run("1", True)
        y_aten, g_aten = run("0", True)
        y_ref, g_ref = run("0", False)
    finally:
        os.environ.pop("DFINE_STEM", None)
        os.environ.pop("DFINE_ALLOW_LIBRARY", None)
        kernels.reload_env()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    assert _rel(y_hip, y_ref) < max(3e-2, 1.5 * _rel(y_aten, y_ref))
    for n in g_ref:
        if g_ref[n].numel() < 8:              # scalar affine parameters: a cosine is just a sign
            continue
        c_hip, c_aten = cos(g_hip[n], g_ref[n]), cos(g_aten[n], g_ref[n])
        assert c_hip > 0.98 and c_hip > c_aten - 0.01, (n, c_hip, c_aten)


@pytest.mark.parametrize("ca,cb,cout,hw", [(24, 24, 24, (64, 128)), (16, 16, 16, (40, 64)), (32, 32, 32, (64, 64))])
def test_stem_conv_two_sources_matches_concatenated_input(cuda, ca, cb, cout, hw):
    """stem3 reading [pooled stem1 | stem2 branch] in place (_StemConv2) is the single-tensor kernel on torch.cat of the two:
    bit-identical forward (same arithmetic order), data gradient returned as two contiguous tensors, same weight gradient."""
    from custom_d_fine_amd import kernels
    torch.manual_seed(ca + cout)
    H, W = hw
    xa = torch.randn(3, ca, H, W, device=cuda).bfloat16().requires_grad_(True)
    xb = torch.randn(3, cb, H, W, device=cuda).bfloat16().requires_grad_(True)
    w = (torch.randn(cout, ca + cb, 3, 3, device=cuda) / ((ca + cb) * 9) ** 0.5).requires_grad_(True)
    y = kernels._StemConv2.apply(xa, xb, w, 1)
    go = torch.randn_like(y)
    y.backward(go)
    xc = torch.cat([xa.detach(), xb.detach()], 1).requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    yr = kernels._StemConv.apply(xc, wr, 2, 1, False)
    yr.backward(go)
    assert torch.equal(y, yr)
    assert xa.grad.is_contiguous() and xb.grad.is_contiguous()
    # the two-tensor data gradient runs on the matrix cores with bf16-rounded weights (csrc/stem3.hip), the single-tensor one
    # on fp32 FMAs: equal to bf16 rounding of the result, not bit for bit
    tol = 2 ** -6 * xc.grad.float().abs().max()
    assert (xa.grad.float() - xc.grad[:, :ca].float()).abs().max() <= tol
    assert (xb.grad.float() - xc.grad[:, ca:].float()).abs().max() <= tol
    assert _rel(w.grad, wr.grad) < 1e-6


@pytest.mark.parametrize("ch,cout,B,hw", [(24, 24, 2, (32, 320)), (24, 24, 1, (320, 320)), (16, 16, 2, (24, 160)), (32, 32, 2, (20, 480)),
                                          (24, 24, 3, (6, 160))])
def test_stem3_row_streaming_kernels_vs_fp32_reference(cuda, ch, cout, B, hw):
    """csrc/stem3.hip (forward: output widths that are multiples of 80; data gradient: multiples of 16) against fp32 conv2d /
    conv_transpose2d on the same bf16 inputs with the weights rounded to bf16 (what the kernels multiply with); results are
    rounded to bf16 once: 2^-7 of the largest magnitude.  Includes image borders, band borders and the 80-column block seams."""
    import torch.nn.functional as F
    from custom_d_fine_amd import hip
    torch.manual_seed(ch + hw[1])
    H, W = hw
    xa = torch.randn(B, ch, H, W, device=cuda).bfloat16()
    xb = torch.randn(B, ch, H, W, device=cuda).bfloat16()
    w = torch.randn(cout, 2 * ch, 3, 3, device=cuda) / (2 * ch * 9) ** 0.5
    wr = w.bfloat16().float()
    y = hip.stem_conv2(xa, xb, hip.stem_pack_weights(w, 0), cout, 3, 2, 1, (H // 2, W // 2))
    yr = F.conv2d(torch.cat([xa, xb], 1).float(), wr, stride=2, padding=1)
    assert y.shape == yr.shape
    assert (y.float() - yr).abs().max() <= 2 ** -7 * yr.abs().max()
    go = torch.randn(B, cout, H // 2, W // 2, device=cuda).bfloat16()
    dxa, dxb = hip.stem_dgrad_s2_2(go, hip.stem_pack_weights(w, 2), ch, ch)
    dr = F.conv_transpose2d(go.float(), wr, stride=2, padding=1, output_padding=1)
    assert dxa.shape == xa.shape and dxb.shape == xb.shape
    tol = 2 ** -7 * dr.abs().max()
    assert (dxa.float() - dr[:, :ch]).abs().max() <= tol and (dxb.float() - dr[:, ch:]).abs().max() <= tol
