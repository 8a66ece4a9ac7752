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
