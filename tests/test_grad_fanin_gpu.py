"""Gradient fan-in without element-wise adds (kernels.GradFanIn): maps with several consumers get their gradient by the consumers'
convolutions adding onto ONE parked buffer in their store epilogues.  Reference: autograd's own adds of the same bf16 gradient
maps (DFINE_FAN_CHAIN=0 / DFINE_GRAD_FANIN=0 run the same kernels and let the engine add) - the sums differ by the order of the
bf16 roundings only (tolerance 2^-6 of the largest gradient element; cosine > 0.9999)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cmp(a, b):
    a, b = a.float(), b.float()
    assert (a - b).abs().max() <= 2 ** -6 * b.abs().max() + 1e-6
    assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0) > 0.9999


@pytest.mark.parametrize("B,H,two_inputs", [(4, 40, True), (2, 80, False), (3, 20, True)])
def test_repncspelan4_chain_matches_autograd_adds(cuda, monkeypatch, B, H, two_inputs):
    """RepNCSPELAN4 (ref hybrid_encoder.py:182-206): cv1's output feeds cv4 whole and cv2's two 1x1 convolutions through its upper
    half, cv2's output feeds cv4 and cv3's two."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hybrid_encoder import RepNCSPELAN4

    from torch.utils._python_dispatch import TorchDispatchMode
    counts = []

    def run(flag):
        monkeypatch.setenv("DFINE_FAN_CHAIN", flag)
        kernels.reload_env()
        torch.manual_seed(1)
        blk = RepNCSPELAN4(512, 256, 512, 128, n=2, act="silu").to(cuda).train()
        xs = [torch.randn(B, 256, H, H, device=cuda).bfloat16().requires_grad_(True) for _ in range(2)]
        if not two_inputs:
            xs = [torch.randn(B, 512, H, H, device=cuda).bfloat16().requires_grad_(True)]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xs if two_inputs else xs[0])
        go = torch.randn(y.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(5)).to(y.dtype)
        adds = []

        class Spy(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                if func.__name__.split(".")[0] in ("add", "add_") and args and torch.is_tensor(args[0]) and args[0].dim() == 4:
                    adds.append(func.__name__)
                return func(*args, **(kwargs or {}))
        with Spy():
            y.backward(go)
        counts.append(len(adds))
        return y, [x.grad for x in xs], {n: p.grad.float().clone() for n, p in blk.named_parameters()}

    try:
        y0, gx0, gp0 = run("0")
        y1, gx1, gp1 = run("1")
    finally:
        monkeypatch.delenv("DFINE_FAN_CHAIN", raising=False)
        kernels.reload_env()
    assert torch.equal(y0, y1)                            # the forward pass is the same launches
    # the engine no longer adds the 256-, 512- and the two 128-channel maps of the block
    assert counts[0] >= 4 and counts[1] == 0, counts
    for a, b in zip(gx1, gx0):
        _cmp(a, b)
    for n in gp0:
        assert gp1[n] is not None
        a, b = gp1[n], gp0[n]
        assert (a - b).abs().max() <= 2e-2 * b.abs().max() + 1e-4, n
        if a.numel() >= 64:
            assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0) > 0.999, n


def test_fan_slice_with_a_consumer_outside_the_chain(cuda):
    """A slice consumer that is not a chain convolution returns its own gradient: autograd adds it to the returned view and
    _FanSlice writes the sum back - nothing is lost."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hybrid_encoder import ConvNormLayer_fuse
    torch.manual_seed(2)
    B, C, H = 2, 128, 40
    conv_a = ConvNormLayer_fuse(64, 64, 1, 1, act="silu").to(cuda).train()
    conv_b = ConvNormLayer_fuse(C + 64, 64, 1, 1, act="silu").to(cuda).train()
    x0 = torch.randn(B, C, H, H, device=cuda).bfloat16()

    def run(chain):
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xx = x * 1.0                                   # a non-leaf map
            if chain:
                s, fan = kernels.fan_slice(xx, 64, 64)
                assert fan is not None
            else:
                s, fan = xx[:, 64:], None
            u = conv_a(s, fanin=fan)
            v = s.float().sin().to(torch.bfloat16)         # the consumer outside the chain
            y = conv_b([xx, u], fans=[fan, None]) + v
        y.float().square().sum().backward()
        return x.grad

    _cmp(run(True), run(False))


def test_take_rows_and_pass_equals_gather_plus_autograd_add(cuda):
    """The decoder's query selection (ref dfine_decoder.py:842-853): rows of the memory by top-k indices next to the value path."""
    from custom_d_fine_amd import kernels
    torch.manual_seed(3)
    B, L, C, K = 4, 2100, 256, 300
    t0 = torch.randn(B, L, C, device=cuda).bfloat16()
    ind = torch.stack([torch.randperm(L, device=cuda)[:K] for _ in range(B)])
    g_all, g_rows = torch.randn(B, L, C, device=cuda).bfloat16(), torch.randn(B, K, C, device=cuda).bfloat16()

    def run(fused):
        t = t0.clone().requires_grad_(True)
        tt = t * 1.0
        if fused:
            full, rows = kernels.take_rows_and_pass(tt, ind)
            assert type(rows.grad_fn).__name__ == "_TakeRowsAndPassBackward"
        else:
            full, rows = tt, tt.gather(1, ind.unsqueeze(-1).expand(-1, -1, C))
        torch.autograd.backward([full * 1.0, rows * 1.0], [g_all.clone(), g_rows.clone()])
        return full.detach(), rows.detach(), t.grad

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # one consumer only: the other gradient is absent
    t = t0.clone().requires_grad_(True)
    full, rows = kernels.take_rows_and_pass(t * 1.0, ind)
    rows.float().sum().backward()
    assert torch.equal(t.grad.float().sum(dim=(1, 2)), torch.full((B,), float(K * C), device=cuda))


@pytest.mark.parametrize("B,C,H,W", [(4, 96, 80, 80), (2, 384, 40, 40), (3, 24, 36, 48)])
def test_depthwise_s2_data_gradient_onto_parked_gradient(cuda, B, C, H, W):
    """HG_Stage.downsample (3x3 / stride 2 depthwise, ref hgnetv2.py:295-303) as the later consumer of a stage output."""
    from custom_d_fine_amd import hip
    torch.manual_seed(C)
    x = torch.randn(B, C, H, W, device=cuda).bfloat16()
    w = torch.randn(C, 1, 3, 3, device=cuda)
    dy = torch.randn(B, C, H // 2, W // 2, device=cuda).bfloat16()
    other = torch.randn_like(x)
    assert hip.dwconv_acc_supported(x, 3, 2, 1)
    dx, _ = hip.dwconv_backward(x, w, dy, 2, 1, True, False)
    got, _ = hip.dwconv_backward(x, w, dy, 2, 1, True, False, acc=other.clone())
    assert torch.equal(got, dx + other)


def test_part_wise_conv_adds_onto_some_output_parts(cuda):
    """dfine_conv1x1_seg_accum_parts_bf16: the aggregation's data gradient of HG_Block, block-input part added onto the residual
    connection's gradient, the other parts overwritten."""
    from custom_d_fine_amd import hip
    torch.manual_seed(9)
    B, H, W, Cin = 3, 40, 40, 96
    chans = (64, 32, 32, 48)
    dy = torch.randn(B, Cin, H, W, device=cuda).bfloat16()
    w = torch.randn(Cin, sum(chans), 1, 1, device=cuda) * Cin ** -0.5       # forward conv: sum(chans) -> Cin
    w2 = hip.conv_pack_weights(w, True)
    ref = [torch.empty(B, c, H, W, device=cuda, dtype=torch.bfloat16) for c in chans]
    hip.conv1x1_seg_forward((dy,), w2, ref)
    for flags in ([True, False, False, False], [False, True, False, True], [True, True, True, True]):
        old = [torch.randn_like(r) for r in ref]
        outs = [o.clone() if f else torch.full_like(o, float("nan")) for o, f in zip(old, flags)]
        hip.conv1x1_seg_forward((dy,), w2, outs, accum=flags)
        for o, r, p, f in zip(outs, ref, old, flags):
            assert torch.equal(o, r + p if f else r)


def _run_blocks(cuda, monkeypatch, park, build, x0, n_out):
    from custom_d_fine_amd import kernels
    monkeypatch.setenv("DFINE_PARK_EAGER", park)
    kernels.reload_env()
    torch.manual_seed(3)
    mods = build()
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = mods(x * 1.0)
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    assert len(outs) == n_out
    gen = torch.Generator(device=cuda).manual_seed(5)
    torch.autograd.backward(list(outs), [torch.randn(o.shape, device=cuda, generator=gen).to(o.dtype) for o in outs])
    return [o.detach() for o in outs], x.grad, {n: p.grad.float().clone() for n, p in mods.named_parameters()}


@pytest.mark.parametrize("light", [True, False])
def test_hg_block_residual_gradient_hand_off(cuda, monkeypatch, light):
    """HG_Block with the residual connection (ref hgnetv2.py:265-275): the block input has three consumers.  Captured segments
    park the connection's gradient and add the aggregation's and layer 0's data gradients onto it in place (DFINE_PARK_EAGER=1
    runs that in an eager pass); the default eager pass lets autograd add.  Three-term bf16 sums in another order."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block
    x0 = torch.randn(4, 128, 40, 40, device=cuda).bfloat16()
    build = lambda: HG_Block(128, 32, 128, 3, residual=True, kernel_size=5 if light else 3, light_block=light, use_lab=True,
                             agg="se").to(cuda).train()
    try:
        y0, gx0, gp0 = _run_blocks(cuda, monkeypatch, "0", build, x0, 1)
        y1, gx1, gp1 = _run_blocks(cuda, monkeypatch, "1", build, x0, 1)
    finally:
        monkeypatch.delenv("DFINE_PARK_EAGER", raising=False)
        kernels.reload_env()
    assert torch.equal(y0[0], y1[0])
    assert (gx0.float() - gx1.float()).abs().max() <= 2 ** -6 * gx0.float().abs().max()
    for n in gp0:                       # nothing upstream of the block input is inside the block: the parameters see the same terms
        assert (gp0[n] - gp1[n]).abs().max() <= 1e-3 * gp0[n].abs().max() + 1e-6, n


def test_stage_output_gradient_hand_off(cuda, monkeypatch):
    """A stage output that leaves the backbone and feeds the next stage's depthwise stride-2 convolution (ref hgnetv2.py:295-303,
    520-526): the outside gradient is parked and the depthwise data gradient is added onto it in place - the same two-term sum
    autograd forms, bit for bit; upstream of it only the float atomics of the weight-gradient kernels differ."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Stage

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = HG_Stage(64, 32, 128, 1, 3, downsample=False, light_block=False, kernel_size=3, use_lab=True, agg="se")
            self.b = HG_Stage(128, 32, 256, 2, 3, downsample=True, light_block=True, kernel_size=5, use_lab=True, agg="se")

        def forward(self, x):
            u = self.a(x)
            fan = kernels.GradFanIn() if kernels.grad_fanin_enabled(u) else None
            v = self.b(u, fanin=fan)
            return [kernels.park_grad(u, fan), v]

    x0 = torch.randn(4, 64, 40, 40, device=cuda).bfloat16()
    build = lambda: Two().to(cuda).train()
    try:
        y0, gx0, gp0 = _run_blocks(cuda, monkeypatch, "0", build, x0, 2)
        y1, gx1, gp1 = _run_blocks(cuda, monkeypatch, "1", build, x0, 2)
    finally:
        monkeypatch.delenv("DFINE_PARK_EAGER", raising=False)
        kernels.reload_env()
    for a, b in zip(y0, y1):
        assert torch.equal(a, b)
    # stage b has a residual block: three-term sums in another order downstream of the stage output
    assert torch.nn.functional.cosine_similarity(gx0.float().flatten(), gx1.float().flatten(), dim=0) > 0.9995
    # (convolution weights only: a BatchNorm scale / shift in front of a depthwise convolution + BatchNorm has an exactly zero
    # gradient - what the kernels produce for those parameters is rounding noise, in either order)
    for n in gp0:
        if n.endswith("conv.weight"):
            assert torch.nn.functional.cosine_similarity(gp0[n].flatten(), gp1[n].flatten(), dim=0) > 0.999, n


@pytest.mark.parametrize("light", [True, False])
def test_hg_block_residual_in_the_batchnorm_apply_pass(cuda, monkeypatch, light):
    """The residual connection of HG_Block (ref hgnetv2.py:274-275) added by the excitation unit's BatchNorm apply kernel
    (dfine_bn_residual_once) against the separate add (DFINE_BN_RESIDUAL=0): the unit's output is rounded to bf16 before the
    add, so outputs and gradients are bit-identical."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch.hgnetv2 import HG_Block
    x0 = torch.randn(4, 128, 40, 40, device=cuda).bfloat16()
    build = lambda: HG_Block(128, 32, 128, 3, residual=True, kernel_size=5 if light else 3, light_block=light, use_lab=True,
                             agg="se").to(cuda).train()
    try:
        monkeypatch.setenv("DFINE_BN_RESIDUAL", "0")
        y0, gx0, gp0 = _run_blocks(cuda, monkeypatch, "0", build, x0, 1)
        monkeypatch.setenv("DFINE_BN_RESIDUAL", "1")
        y1, gx1, gp1 = _run_blocks(cuda, monkeypatch, "0", build, x0, 1)
    finally:
        monkeypatch.delenv("DFINE_BN_RESIDUAL", raising=False)
        monkeypatch.delenv("DFINE_PARK_EAGER", raising=False)
        kernels.reload_env()
    assert torch.equal(y0[0], y1[0])
    assert torch.equal(gx0, gx1)
    for n in gp0:
        assert (gp0[n] - gp1[n]).abs().max() <= 1e-3 * gp0[n].abs().max() + 1e-6, n


def test_fan_out_sums_the_consumers_gradients_in_one_pass(cuda):
    """kernels.fan_out (decoder token streams, ref dfine_decoder.py:214-255): k aliases, one fused fp32 sum in backward - the same
    gradient as autograd's pairwise adds up to the order of the fp32 additions (exact for three terms of this test's magnitudes
    is not guaranteed: tolerance 1e-6 relative)."""
    from custom_d_fine_amd import kernels
    torch.manual_seed(7)
    x0 = torch.randn(8, 492, 256, device=cuda)
    gs = [torch.randn_like(x0) for _ in range(5)]

    def run(fused):
        x = x0.clone().requires_grad_(True)
        y = x * 1.0
        taps = kernels.fan_out(y, 5) if fused else (y,) * 5
        if fused:
            assert type(taps[0].grad_fn).__name__ == "_FanOutBackward"
        outs = [taps[0] * 2.0, taps[1].sin(), taps[2] + 1.0, taps[4] * taps[4]]          # alias 3 stays unused
        torch.autograd.backward(outs, gs[:4])
        return x.grad

    a, b = run(True), run(False)
    assert (a - b).abs().max() <= 1e-6 * b.abs().max()
