"""A1/A2 parity: HIP depthwise convolution and fused BatchNorm+activation+affine vs a plain PyTorch
fp32 reference of the same op (float kernels: tolerance 1e-4 fp32, 3e-2 bf16 storage)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch.hgnetv2 import ConvBNAct, LightConvBNAct

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C,H,W,K,S", [(3, 16, 40, 40, 5, 1), (2, 24, 81, 79, 3, 2), (2, 8, 160, 160, 3, 2),
                                          (1, 5, 7, 9, 5, 1), (2, 6, 20, 20, 3, 1)])
def test_depthwise_conv_fwd_bwd(cuda, B, C, H, W, K, S):
    torch.manual_seed(K * 10 + S)
    x = torch.randn(B, C, H, W)
    w = torch.randn(C, 1, K, K) * 0.3
    P = (K - 1) // 2
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=S, padding=P, groups=C)
    go = torch.randn_like(yr)
    yr.backward(go)
    xg, wg = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True)
    y = kernels._DepthwiseConv.apply(xg, wg, S, P)
    y.backward(go.to(cuda))
    assert torch.allclose(y.cpu(), yr, rtol=1e-4, atol=1e-4)
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(wg.grad.cpu(), wr.grad, rtol=1e-3, atol=1e-3 * wr.grad.abs().max().item())


@pytest.mark.parametrize("B,C,H,W,K,S", [(3, 16, 40, 40, 5, 1), (2, 8, 20, 20, 5, 1), (2, 8, 80, 80, 3, 1),
                                          (2, 6, 24, 12, 3, 1), (2, 8, 160, 160, 3, 2), (3, 12, 40, 40, 3, 2),
                                          (2, 6, 20, 24, 3, 2), (2, 4, 80, 72, 3, 2)])
def test_depthwise_conv_bf16_vector_kernels(cuda, B, C, H, W, K, S):
    """The register-tiled bf16 kernels (stride 1: 16 / 8-byte strips; 3x3 stride 2) against fp32 conv2d on the
    same bf16-rounded inputs; outputs are rounded to bf16 once, gradients of the weights stay fp32."""
    torch.manual_seed(K * 10 + S)
    x = torch.randn(B, C, H, W).to(torch.bfloat16)
    w = torch.randn(C, 1, K, K) * 0.3
    P = (K - 1) // 2
    xr, wr = x.float().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=S, padding=P, groups=C)
    go = torch.randn_like(yr).to(torch.bfloat16)
    yr.backward(go.float())
    xg, wg = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True)
    y = kernels._DepthwiseConv.apply(xg, wg, S, P)
    assert y.dtype == torch.bfloat16
    y.backward(go.to(cuda))
    tol = 2 ** -7
    assert (y.float().cpu() - yr).abs().max() <= tol * yr.abs().max()
    assert (xg.grad.float().cpu() - xr.grad).abs().max() <= tol * xr.grad.abs().max()
    assert torch.allclose(wg.grad.cpu(), wr.grad, rtol=1e-3, atol=1e-3 * wr.grad.abs().max().item())


@pytest.mark.parametrize("act", [None, "relu", "silu"])
@pytest.mark.parametrize("shape", [(4, 12, 20, 20), (2, 7, 9, 11), (3, 32, 80, 80),
                                   (4, 64, 20, 20), (16, 72, 40, 40), (5, 65, 6, 10)])   # the last three: one block per channel (fp32)
@pytest.mark.parametrize("lab", [False, True])
def test_bn_act_train_fwd_bwd(cuda, act, shape, lab):
    torch.manual_seed(0)
    B, C, H, W = shape
    x = torch.randn(B, C, H, W) * 2 + 0.5
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    ls = torch.tensor([1.3]); lb = torch.tensor([-0.2])
    f = {None: lambda t: t, "relu": F.relu, "silu": F.silu}[act]

    def ref(xin, g, b, s, t, rm, rv, training):
        y = f(F.batch_norm(xin, rm, rv, g, b, training, 0.1, 1e-5))
        return s * y + t if lab else y

    xr = x.clone().requires_grad_(True)
    g, b = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    s, t = ls.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    yr = ref(xr, g, b, s, t, rm, rv, True)
    go = torch.randn_like(yr)
    yr.backward(go)

    xg = x.to(cuda).requires_grad_(True)
    gg, bg = bn.weight.detach().to(cuda).requires_grad_(True), bn.bias.detach().to(cuda).requires_grad_(True)
    sg, tg = ls.to(cuda).requires_grad_(True), lb.to(cuda).requires_grad_(True)
    rmg, rvg = bn.running_mean.clone().to(cuda), bn.running_var.clone().to(cuda)
    y = kernels._BNAct.apply(xg, gg, bg, sg if lab else None, tg if lab else None, rmg, rvg, act, True, 0.1, 1e-5)
    y.backward(go.to(cuda))
    assert torch.allclose(y.cpu(), yr, rtol=1e-4, atol=1e-4)
    assert torch.allclose(rmg.cpu(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(rvg.cpu(), rv, rtol=1e-5, atol=1e-6)
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(gg.grad.cpu(), g.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(bg.grad.cpu(), b.grad, rtol=1e-3, atol=1e-3)
    if lab:
        assert torch.allclose(sg.grad.cpu(), s.grad, rtol=1e-3, atol=1e-2)
        assert torch.allclose(tg.grad.cpu(), t.grad, rtol=1e-3, atol=1e-2)


def test_bn_act_eval_mode(cuda):
    torch.manual_seed(1)
    x = torch.randn(2, 6, 10, 10)
    rm, rv = torch.randn(6) * 0.2, torch.rand(6) + 0.5
    g, b = torch.rand(6) + 0.5, torch.randn(6) * 0.1
    xr = x.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rm, rv, g, b, False, 0.1, 1e-5))
    go = torch.randn_like(yr)
    yr.backward(go)
    xg = x.to(cuda).requires_grad_(True)
    rmg, rvg = rm.to(cuda), rv.to(cuda)
    y = kernels._BNAct.apply(xg, g.to(cuda), b.to(cuda), None, None, rmg, rvg, "relu", False, 0.1, 1e-5)
    y.backward(go.to(cuda))
    assert torch.allclose(y.cpu(), yr, rtol=1e-4, atol=1e-5)
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    assert torch.equal(rmg.cpu(), rm) and torch.equal(rvg.cpu(), rv)          # untouched in eval


def test_units_match_aten_composition_bf16(cuda):
    """Whole ConvBNAct / LightConvBNAct units (HIP tail, HIP depthwise) vs the same modules on CPU fp32."""
    torch.manual_seed(2)
    for unit in (ConvBNAct(16, 24, 3, use_lab=True), LightConvBNAct(16, 32, 5, use_lab=True),
                 ConvBNAct(24, 24, 3, stride=2, groups=24, use_act=False, use_lab=True)):
        x = torch.randn(4, unit.conv1.conv.in_channels if hasattr(unit, "conv1") else unit.conv.in_channels, 40, 40)
        unit.train()
        ref = unit(x)
        import copy
        g = copy.deepcopy(unit).to(cuda)
        for dt, tol in ((torch.float32, 2e-3), (torch.bfloat16, 6e-2)):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
                out = g(x.to(cuda))
            assert out.dtype == dt
            assert (out.float().cpu() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,C,H,W,act,with_res", [(4, 128, 40, 40, "silu", True), (2, 64, 80, 80, "silu", False),
                                                   (3, 32, 20, 20, "relu", True), (2, 16, 24, 8, None, False)])
def test_repvgg_unit_fused_vs_fp32_reference(cuda, B, C, H, W, act, with_res):
    """act(BN(conv3x3(x)) + BN(conv1x1(x))) [+ residual] through the one-pass RepVGG kernels (dfine_bn2_act_*) against a
    plain fp32 PyTorch composition on the same bf16-rounded input and weights; tolerance = bf16 storage of the conv outputs
    (the two convolutions themselves are the MFMA kernels, checked separately in test_conv_mfma_gpu.py)."""
    from custom_d_fine_amd.d_fine.arch.hybrid_encoder import VGGBlock
    torch.manual_seed(C + H)
    act_mod = {"silu": nn.SiLU(), "relu": nn.ReLU(), None: None}[act]
    blk = VGGBlock(C, C, act=act_mod).to(cuda).train()
    with torch.no_grad():
        for bn in (blk.conv1.norm, blk.conv2.norm):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        for conv in (blk.conv1.conv, blk.conv2.conv):
            conv.weight.copy_(conv.weight.bfloat16().float())
    ref = VGGBlock(C, C, act=act_mod).to(cuda).train()
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(B, C, H, W, device=cuda).bfloat16()
    res = torch.randn(B, C, H, W, device=cuda).bfloat16() if with_res else None
    go = torch.randn(B, C, H, W, device=cuda)

    xg = x.clone().requires_grad_(True)
    rg = None if res is None else res.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(xg, residual=rg)
    assert y.dtype == torch.bfloat16
    y.float().mul(go).sum().backward()

    # fp32 reference (plain modules, no autocast)
    xr = x.float().requires_grad_(True)
    rr = None if res is None else res.float().requires_grad_(True)
    def stored(c):                    # the conv outputs live in bf16 (straight-through rounding keeps the graph fp32)
        return c + (c.bfloat16().float() - c).detach()

    z = ref.conv1.norm(stored(ref.conv1.conv(xr))) + ref.conv2.norm(stored(ref.conv2.conv(xr)))
    yr = z if act_mod is None else act_mod(z)
    if rr is not None:
        yr = yr + rr
    yr.mul(go).sum().backward()

    def close(a, b, tol):
        return (a.float() - b.float()).abs().max().item() <= tol * max(b.float().abs().max().item(), 1.0)

    def close_l2(a, b, tol):          # gradients: ReLU masks flip where bf16 rounding moves z across 0, so compare in norm
        return (a.float() - b.float()).norm().item() <= tol * b.float().norm().item()

    assert close(y, yr, 2e-2)
    assert close_l2(xg.grad, xr.grad, 3e-2)
    if rg is not None:
        assert close(rg.grad, rr.grad, 1e-2)
    for name in ("conv1.norm.weight", "conv1.norm.bias", "conv2.norm.weight", "conv2.norm.bias",
                 "conv1.conv.weight", "conv2.conv.weight"):
        a, b = dict(blk.named_parameters())[name].grad, dict(ref.named_parameters())[name].grad
        assert a is not None and close_l2(a, b, 3e-2), name
    for name in ("conv1.norm.running_mean", "conv1.norm.running_var", "conv2.norm.running_mean", "conv2.norm.running_var"):
        a, b = dict(blk.named_buffers())[name], dict(ref.named_buffers())[name]
        assert close(a, b, 1e-2), name
    assert int(blk.conv1.norm.num_batches_tracked) == 1 and int(blk.conv2.norm.num_batches_tracked) == 1


@pytest.mark.parametrize("lab", [False, True])
def test_frozen_batchnorm_unit_takes_hip_tail(cuda, lab):
    """ConvBNAct with a FrozenBatchNorm2d (backbone of D-FINE-l / x: statistics and affine are buffers) under bf16 autocast: the
    fused HIP tail in eval mode - no ATen mul / add passes - against the same module on the CPU in fp32, forward and input
    gradient; the buffers receive no gradient and stay untouched."""
    from custom_d_fine_amd.d_fine.arch.common import FrozenBatchNorm2d, freeze_batch_norm2d
    torch.manual_seed(3)
    unit = ConvBNAct(32, 64, 1, use_act=True, use_lab=lab)
    with torch.no_grad():
        unit.bn.weight.uniform_(0.5, 1.5); unit.bn.bias.normal_(0, 0.3)
        unit.bn.running_mean.normal_(0, 0.2); unit.bn.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in unit.bn.state_dict().items() if k != "num_batches_tracked"}
    freeze_batch_norm2d(unit)
    assert isinstance(unit.bn, FrozenBatchNorm2d)
    unit.bn.load_state_dict(state)
    unit.train()
    x = torch.randn(4, 32, 40, 40)
    go = torch.randn(4, 64, 40, 40)
    xr = x.clone().requires_grad_(True)
    yr = unit(xr)
    yr.backward(go)
    g = unit.to(cuda)
    xg = x.to(cuda).requires_grad_(True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = g(xg)
        y.backward(go.to(cuda).to(y.dtype))
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("bn_apply" in n or "bn_one" in n for n in names), names
    if not lab:                                                  # (the learnable affine of a unit still wants its two sums)
        assert not any("bn_bwd_reduce" in n for n in names), "a frozen unit's backward needs no reduction pass"
    assert (y.float().cpu() - yr).abs().max() <= 3e-2 * yr.abs().max()
    # (a ReLU whose bf16 pre-activation lands on the other side of zero flips single elements: the gradient is compared in norm)
    assert (xg.grad.cpu() - xr.grad).norm() <= 6e-2 * xr.grad.norm()
    for k, v in state.items():
        assert torch.equal(g.bn.state_dict()[k].cpu(), v)


@pytest.mark.parametrize("cin,cout,k,H,W,act,use_lab", [(128, 128, 1, 40, 40, "relu", False), (96, 192, 1, 80, 80, None, True),
                                                        (64, 64, 3, 40, 40, "relu", True), (128, 256, 3, 20, 20, "silu", False),
                                                        (256, 80, 1, 20, 20, "silu", False)])
def test_eval_unit_is_one_launch_with_the_affine_epilogue(cuda, monkeypatch, cin, cout, k, H, W, act, use_lab):
    """Inference (no_grad, eval-mode BatchNorm): conv -> BN -> act [-> LAB] runs as the convolution with the affine + activation
    epilogue (dfine_conv_affine_once) - against the fp32 ATen composition of the same unit, and against the two-pass HIP form
    (conv, then the BatchNorm / activation kernel) it replaces."""
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import LearnableAffineBlock
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False).to(cuda)
    bn = nn.BatchNorm2d(cout).to(cuda)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    lab = LearnableAffineBlock(1.3, -0.2).to(cuda) if use_lab else None
    x = torch.randn(4, cin, H, W, device=cuda)
    calls = []
    real = hip.conv_forward_affine
    monkeypatch.setattr(hip, "conv_forward_affine", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.conv_bn_act(x, conv, bn, act, lab)
        monkeypatch.setattr(kernels, "EVAL_EPILOGUE", False)
        y2 = kernels.conv_bn_act(x, conv, bn, act, lab)
    assert len(calls) == 1 and y.dtype == torch.bfloat16
    f = {None: lambda t: t, "relu": F.relu, "silu": F.silu}[act]
    with torch.no_grad():
        want = f(bn(F.conv2d(x.bfloat16().float(), conv.weight.bfloat16().float(), None, 1, k // 2)))
        want = lab(want) if lab is not None else want
    scale = want.abs().max().item()
    assert (y.float() - want).abs().max().item() <= 1.2e-2 * scale              # one bf16 rounding of the result
    assert (y2.float() - want).abs().max().item() <= 2.5e-2 * scale             # the two-pass form rounds the convolution first
    assert (y.float() - y2.float()).abs().max().item() <= 2.5e-2 * scale


def test_deployed_unit_bias_act_in_the_store_phase(cuda, monkeypatch):
    from custom_d_fine_amd import hip
    torch.manual_seed(3)
    conv = nn.Conv2d(128, 128, 3, 1, 1, bias=True).to(cuda)
    x = torch.randn(2, 128, 40, 40, device=cuda)
    calls = []
    real = hip.conv_forward_affine
    monkeypatch.setattr(hip, "conv_forward_affine", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.conv_bias_act(x, conv, "silu")
    with torch.no_grad():
        want = F.silu(F.conv2d(x.bfloat16().float(), conv.weight.bfloat16().float(), conv.bias, 1, 1))
    assert len(calls) == 1
    assert (y.float() - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()


@pytest.mark.parametrize("C,k,s,H,W,act,use_lab", [(128, 5, 1, 40, 40, "relu", True), (256, 3, 2, 40, 40, None, False), (96, 3, 1, 20, 20, "silu", False)])
def test_eval_depthwise_unit_is_one_launch(cuda, monkeypatch, C, k, s, H, W, act, use_lab):
    """Inference: depthwise conv -> eval-mode BatchNorm -> act [-> LAB] as the depthwise kernel with its affine epilogue
    (dfine_dwconv_affine_once), against the fp32 ATen composition on the bf16-rounded operands."""
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import LearnableAffineBlock
    torch.manual_seed(C + k)
    conv = nn.Conv2d(C, C, k, s, k // 2, groups=C, bias=False).to(cuda)
    bn = nn.BatchNorm2d(C).to(cuda)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    lab = LearnableAffineBlock(0.8, 0.1).to(cuda) if use_lab else None
    x = torch.randn(4, C, H, W, device=cuda).bfloat16()
    calls = []
    real = hip.dwconv_forward_affine
    monkeypatch.setattr(hip, "dwconv_forward_affine", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.conv_bn_act(x, conv, bn, act, lab)
    assert len(calls) == 1 and y.dtype == torch.bfloat16
    f = {None: lambda t: t, "relu": F.relu, "silu": F.silu}[act]
    with torch.no_grad():
        want = f(bn(F.conv2d(x.float(), conv.weight.float(), None, s, k // 2, groups=C)))
        want = lab(want) if lab is not None else want
    assert (y.float() - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()


def test_frozen_eval_folds_follow_a_weight_reload(cuda):
    """freeze_eval_affine keeps the BatchNorm folds on the modules (Torch_model: one launch per unit); loading other weights
    afterwards must not serve the old folds."""
    torch.manual_seed(9)
    conv = nn.Conv2d(64, 64, 1, bias=False).to(cuda)
    bn = nn.BatchNorm2d(64).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    unit = nn.Sequential(conv, bn)
    assert kernels.freeze_eval_affine(unit) == 1 and "_dfine_fold" in bn.__dict__
    x = torch.randn(2, 64, 20, 20, device=cuda)

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return kernels.conv_bn_act(x, conv, bn, "relu", None).float()

    def ref():
        with torch.no_grad():
            return F.relu(bn(F.conv2d(x.bfloat16().float(), conv.weight.bfloat16().float())))

    assert (run() - ref()).abs().max() <= 1.2e-2 * ref().abs().max()
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    state["weight"] = state["weight"] * 3.0
    state["running_mean"] = state["running_mean"] + 1.0
    bn.load_state_dict(state)
    assert (run() - ref()).abs().max() <= 1.2e-2 * ref().abs().max()          # the stale fold would be off by the new scale / shift
    assert "_dfine_fold" not in bn.__dict__


def test_deployed_1x1_reads_a_concatenation_as_parts(cuda, monkeypatch):
    """A deployed 1x1 layer on a list input (the FPN / PAN concatenations in front of the CSP layers): inference takes the part-wise
    kernel with the bias + activation epilogue - no torch.cat - and matches the fp32 composition on the concatenated map."""
    from custom_d_fine_amd import hip
    torch.manual_seed(11)
    conv = nn.Conv2d(256 + 128, 192, 1, bias=True).to(cuda)
    a, b = torch.randn(2, 256, 40, 40, device=cuda).bfloat16(), torch.randn(2, 128, 40, 40, device=cuda).bfloat16()
    cats = []
    real_cat = torch.cat
    monkeypatch.setattr(torch, "cat", lambda *ar, **kw: (cats.append(1), real_cat(*ar, **kw))[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.conv_bias_act([a, b], conv, "silu")
    assert not cats and y.dtype == torch.bfloat16
    with torch.no_grad():
        want = F.silu(F.conv2d(real_cat([a, b], 1).float(), conv.weight.bfloat16().float(), conv.bias))
    assert (y.float() - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()
