"""A7 parity: HIP deformable-attention gather (through the C ABI) vs the oracle and the golden
vectors from the reference.  fp32 tolerance 1e-5 (fwd) / 1e-4 (bwd, atomics reorder sums);
bf16 storage tolerance 2e-2 relative to the output scale."""
import numpy as np
import pytest
import torch

from custom_d_fine_amd import kernels
from oracle import np_ref
from tests import helpers

pytestmark = pytest.mark.gpu
G = helpers.GOLDEN_DIR


def _run(value, loc, w, go, shapes, points, dev, dtype=torch.float32):
    v = torch.tensor(value, device=dev, dtype=dtype, requires_grad=True)
    lc = torch.tensor(loc, device=dev, requires_grad=True)
    ww = torch.tensor(w, device=dev, requires_grad=True)
    out = kernels.msda(v, shapes, lc, ww, points)
    out.backward(torch.tensor(go, device=dev, dtype=dtype))
    return out.detach().float().cpu().numpy(), v.grad.float().cpu().numpy(), lc.grad.cpu().numpy(), ww.grad.cpu().numpy()


def test_golden_case_d16(cuda):
    g = np.load(f"{G}/msda.npz")
    value, loc, w, go, shapes, points = helpers.make_msda_case(1, B=1, Lq=5, H=2, D=16, shapes=((5, 7), (3, 2)), points=(2, 4))
    out, gv, gl, gw = _run(value, loc, w, go, shapes, points, cuda)
    np.testing.assert_allclose(out, g["s1/out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gv, g["s1/g_value"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gw, g["s1/g_weight"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gl, g["s1/g_loc"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("D,H,B,Lq", [(32, 8, 2, 37), (16, 8, 3, 70), (64, 4, 1, 9), (32, 5, 2, 33)])
def test_vs_oracle_fp32_incl_edges(cuda, D, H, B, Lq):
    value, loc, w, go, shapes, points = helpers.make_msda_case(D + H, B=B, Lq=Lq, H=H, D=D)
    out, gv, gl, gw = _run(value, loc, w, go, shapes, points, cuda)
    np.testing.assert_allclose(out, np_ref.msda_forward(value, shapes, loc, w, points), rtol=1e-5, atol=1e-5)
    rv, rl, rw = np_ref.msda_backward(value, shapes, loc, w, points, go)
    np.testing.assert_allclose(gv, rv, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gw, rw, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gl, rl, rtol=1e-4, atol=3e-4)


def test_fused_matches_reference_module_golden(cuda):
    g = np.load(f"{G}/msda.npz")
    # golden module case has head_dim 4; the kernels need 16/32/64 -> tile the channels x4
    shapes, points = ((8, 8), (4, 4), (2, 2)), (3, 6, 3)
    value = np.tile(g["mod/value"], (1, 1, 1, 4))                       # [B, L, 8, 16]
    go = np.tile(g["mod/grad_out"].reshape(2, 9, 8, 4), (1, 1, 1, 4)).reshape(2, 9, 128)
    v = torch.tensor(value, device=cuda, requires_grad=True)
    off = torch.tensor(g["mod/offsets"], device=cuda, requires_grad=True)
    lg = torch.tensor(g["mod/logits"], device=cuda, requires_grad=True)
    out = kernels.msda_fused(v, shapes, torch.tensor(g["mod/ref"], device=cuda), off, lg, points, 0.5)
    out.backward(torch.tensor(go, device=cuda))
    want = np.tile(g["mod/out"].reshape(2, 9, 8, 4), (1, 1, 1, 4)).reshape(2, 9, 128)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(v.grad.cpu().numpy(), np.tile(g["mod/g_value"], (1, 1, 1, 4)), rtol=1e-4, atol=1e-5)
    # 4 identical channel groups -> 4x the gradient on the shared offsets / logits
    np.testing.assert_allclose(off.grad.cpu().numpy(), 4 * g["mod/g_offsets"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), 4 * g["mod/g_logits"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_vs_oracle(cuda, dtype):
    rng = np.random.default_rng(3)
    B, Lq, H, D = 2, 45, 8, 32
    shapes, points = ((10, 12), (5, 6), (3, 3)), (3, 6, 3)
    L, P = 120 + 30 + 9, 12
    value = rng.normal(0, 1, (B, L, H, D)).astype(np.float32)
    ref = np.concatenate([rng.uniform(0, 1, (B, Lq, 2)), rng.uniform(0.02, 0.6, (B, Lq, 2))], -1).astype(np.float32)
    off = rng.normal(0, 2, (B, Lq, H, P, 2)).astype(np.float32)
    lg = rng.normal(0, 2, (B, Lq, H, P)).astype(np.float32)
    go = rng.normal(0, 1, (B, Lq, H * D)).astype(np.float32)
    tv = torch.tensor(value, device=cuda, dtype=dtype, requires_grad=True)
    to = torch.tensor(off, device=cuda, dtype=dtype, requires_grad=True)
    tl = torch.tensor(lg, device=cuda, dtype=dtype, requires_grad=True)
    out = kernels.msda_fused(tv, shapes, torch.tensor(ref, device=cuda), to, tl, points, 0.5)
    out.backward(torch.tensor(go, device=cuda, dtype=dtype))
    # oracle on the values the kernel actually saw (bf16-rounded inputs)
    v_in, o_in, l_in = (t.detach().float().cpu().numpy() for t in (tv, to, tl))
    go_in = torch.tensor(go).to(dtype).float().numpy()
    loc, w = np_ref.msda_prologue(ref, o_in, l_in, points, 0.5)
    want = np_ref.msda_forward(v_in, shapes, loc, w, points)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), want, **tol)
    rv, rl, rw = np_ref.msda_backward(v_in, shapes, loc, w, points, go_in)
    scale = np.array([1 / n for n in points for _ in range(n)], np.float32)
    g_off = rl * scale[None, None, None, :, None] * ref[:, :, None, None, 2:] * 0.5
    g_log = w * (rw - (w * rw).sum(-1, keepdims=True))
    btol = dict(rtol=1e-4, atol=3e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(tv.grad.float().cpu().numpy(), rv, **btol)
    np.testing.assert_allclose(to.grad.float().cpu().numpy(), g_off, **btol)
    np.testing.assert_allclose(tl.grad.float().cpu().numpy(), g_log, **btol)


def test_empty_and_ragged(cuda):
    shapes, points = ((4, 4), (2, 2)), (2, 2)
    v = torch.randn(2, 20, 8, 32, device=cuda)
    out = kernels.msda(v, shapes, torch.zeros(2, 0, 8, 4, 2, device=cuda), torch.zeros(2, 0, 8, 4, device=cuda), points)
    assert out.shape == (2, 0, 256)
    # a query count that is not a multiple of the per-block task count
    loc = torch.rand(2, 33, 8, 4, 2, device=cuda)
    w = torch.softmax(torch.randn(2, 33, 8, 4, device=cuda), -1)
    out = kernels.msda(v, shapes, loc, w, points)
    want = np_ref.msda_forward(v.cpu().numpy(), shapes, loc.cpu().numpy(), w.cpu().numpy(), points)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):    # level table that does not cover the value rows
        kernels.msda(v, ((4, 4),), loc[:, :, :, :2], w[:, :, :, :2], (2,))


def test_full_size_properties_bf16(cuda):
    """D-FINE-m / 640x640 / bs=32 shapes (BASELINE.json configs[2]): size-independent properties
    - constant field stays constant for in-range samples, linearity in value, and the adjoint
    identity <grad_out, A v> == <A^T grad_out, v> that ties backward to forward."""
    torch.manual_seed(0)
    B, Lq, H, D = 32, 496, 8, 32
    shapes, points = ((80, 80), (40, 40), (20, 20)), (3, 6, 3)
    L = 8400
    ref = torch.cat([torch.rand(B, Lq, 2, device=cuda) * 0.6 + 0.2, torch.rand(B, Lq, 2, device=cuda) * 0.1], -1)
    off = torch.randn(B, Lq, H, 12, 2, device=cuda).clamp(-2, 2)
    lg = torch.randn(B, Lq, H, 12, device=cuda)
    const = torch.full((B, L, H, D), 0.75, device=cuda, dtype=torch.bfloat16)
    out = kernels.msda_fused(const, shapes, ref, off.bfloat16(), lg.bfloat16(), points, 0.5)
    assert (out.float() - 0.75).abs().max() < 8e-3          # all samples in range -> weights sum to 1
    v1 = torch.randn(B, L, H, D, device=cuda)
    v2 = torch.randn(B, L, H, D, device=cuda)
    f = lambda v: kernels.msda_fused(v, shapes, ref, off, lg, points, 0.5)
    lin = f(2 * v1 - 3 * v2) - (2 * f(v1) - 3 * f(v2))
    assert lin.abs().max() < 1e-4
    v = v1.clone().requires_grad_(True)
    go = torch.randn(B, Lq, H * D, device=cuda)
    o = kernels.msda_fused(v, shapes, ref, off, lg, points, 0.5)
    o.backward(go)
    lhs = (go.double() * o.detach().double()).sum()
    rhs = (v.grad.double() * v.detach().double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-5
    # spot-check one image against the oracle
    loc, w = np_ref.msda_prologue(ref[:1].cpu().numpy(), off[:1].cpu().numpy(), lg[:1].cpu().numpy(), points, 0.5)
    want = np_ref.msda_forward(v1[:1].cpu().numpy(), shapes, loc, w, points)
    np.testing.assert_allclose(o[:1].detach().cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_shared_value_gradient_accumulator(cuda):
    """Several gathers from ONE value tensor (the decoder layers): with msda_share_value_grad their backward passes add into a
    single fp32 accumulator and the total is returned once - the same gradient as the per-call path summed by autograd."""
    torch.manual_seed(5)
    B, Lq, H, D = 2, 40, 8, 32
    shapes, points = ((10, 12), (5, 6), (3, 3)), (3, 6, 3)
    L = 120 + 30 + 9
    ref = torch.cat([torch.rand(B, Lq, 2, device=cuda), torch.rand(B, Lq, 2, device=cuda) * 0.5 + 0.02], -1)
    offs = [torch.randn(B, Lq, H, 12, 2, device=cuda).bfloat16() for _ in range(3)]
    lgs = [torch.randn(B, Lq, H, 12, device=cuda).bfloat16() for _ in range(3)]
    gos = [torch.randn(B, Lq, H * D, device=cuda).bfloat16() for _ in range(3)]
    base = torch.randn(B, L, H * D, device=cuda).bfloat16()

    def run(shared):
        mem = base.clone().requires_grad_(True)
        value = mem.reshape(B, L, H, D)
        if shared:
            value = kernels.msda_share_value_grad(value)
            assert hasattr(value, "_dfine_share")
        outs = [kernels.msda_fused(value, shapes, ref, o, l, points, 0.5) for o, l in zip(offs, lgs)]
        torch.autograd.backward(outs, gos)
        return mem.grad.float()

    plain, shared = run(False), run(True)
    assert plain.abs().max() > 0
    # one rounding of the fp32 total instead of three roundings + two bf16 adds: equal within bf16 resolution
    assert (plain - shared).abs().max() <= 2e-2 * plain.abs().max()
    # a subset of the uses never reports a total twice or loses the accumulator
    mem = base.clone().requires_grad_(True)
    value = kernels.msda_share_value_grad(mem.reshape(B, L, H, D))
    o1 = kernels.msda_fused(value, shapes, ref, offs[0], lgs[0], points, 0.5)
    o1.backward(gos[0])
    assert mem.grad is not None and torch.isfinite(mem.grad.float()).all()


def test_value_gradient_accumulate_modes(cuda, monkeypatch):
    """d(value) accumulation forms of dfine_msda_fused_bwd_acc at the bench shape (clustered denoising queries + uniform
    ones): scaled f16 with one packed atomic per channel pair (the bf16 default) and int32 fixed point in 64-bit integer
    atomics against the f32-atomic accumulator.  Two calls share each accumulator; the second brings 64x larger gradients,
    which forces the in-place rescale of the scaled forms.  f16: 11 significant bits per running sum - measured 3.6e-3 of the
    largest entry on these clustered queries (hot cells collect hundreds of addends), the size of the bf16 rounding of the result; fixed point: exact sums of contributions rounded to 2^-30 of the overflow
    bound, bit-identical from run to run (integer adds commute; f32 atomics do not)."""
    from custom_d_fine_amd import hip
    torch.manual_seed(1)
    B, Lq, H, D, L = 8, 492, 8, 32, 8400
    shapes, points = ((80, 80), (40, 40), (20, 20)), (3, 6, 3)
    value = torch.randn(B, L, H, D, device=cuda).bfloat16()
    gt = torch.cat([torch.rand(B, 7, 2, device=cuda) * 0.6 + 0.2, torch.rand(B, 7, 2, device=cuda) * 0.3 + 0.05], -1)
    dn = (gt.repeat(1, 28, 1)[:, :192] + torch.randn(B, 192, 4, device=cuda) * 0.02).clamp(0.01, 0.99)
    dn = dn.view(B, 12, 16, 4).clone()
    dn[:, :, 11:] = 1e-5                    # padding entries of the denoising groups: zero boxes, every point on pixel (0, 0)
    dn = dn.view(B, 192, 4)
    ref = torch.cat([dn, torch.cat([torch.rand(B, 300, 2, device=cuda), torch.rand(B, 300, 2, device=cuda) * 0.3 + 0.02], -1)], 1).contiguous()
    off = (torch.randn(B, Lq, H, 12, 2, device=cuda) * 0.5).bfloat16()
    lg = torch.randn(B, Lq, H, 12, device=cuda).bfloat16()
    gos = [torch.randn(B, Lq, H * D, device=cuda).bfloat16(), (torch.randn(B, Lq, H * D, device=cuda) * 64).bfloat16()]

    def run(mode):
        monkeypatch.setattr(hip, "MSDA_ACC_MODE", mode)
        acc = hip.msda_grad_value_buffer(value, uses=2)
        small = []
        for go in gos:
            _, goff, glog = hip.msda_fused_backward(value, ref, off, lg, go, shapes, points, 0.5, gv_acc=acc)
            small.append((goff.float(), glog.float()))
        return hip.msda_finish_grad_value(acc, torch.float32), small

    want, small0 = run(0)
    top = want.abs().max().item()
    f16, small2 = run(2)
    assert (f16 - want).abs().max().item() <= 5e-3 * top
    assert ((f16 - want).abs().sum() / want.abs().sum()).item() < 1e-3
    fx, small3 = run(3)
    assert (fx - want).abs().max().item() <= 2e-4 * top
    fx2, _ = run(3)
    assert torch.equal(fx, fx2)
    for other in (small2, small3):                    # the per-point gradients do not depend on the accumulator form
        for (a0, b0), (a1, b1) in zip(small0, other): # (two lane mappings: fp32 sums in another order, bf16 results 1 ulp apart)
            assert (a0 - a1).abs().max() <= 1e-2 * a0.abs().max() and (b0 - b1).abs().max() <= 1e-2 * b0.abs().max()
    assert torch.equal(small3[0][0], run(3)[1][0][0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,H,B,Lq", [(32, 8, 1, 23), (16, 4, 2, 45), (64, 2, 2, 7), (32, 3, 3, 41)])
def test_backward_with_coinciding_rows(cuda, dtype, D, H, B, Lq):
    """The backward kernels sum contributions that meet in one value row in registers before the atomic (over the points of a
    level inside a task, over the tasks of a wave at the end of a level).  Rows that coincide on purpose: runs of consecutive
    queries with all points on pixel (0, 0) of every level (the padding entries of the denoising groups), runs that continue
    across an image boundary (same row index, different image: must NOT merge), identical neighbours at other pixels, a last
    workgroup that is only partly filled, row 0 of image 0 with B = 1."""
    value, loc, w, go, shapes, points = helpers.make_msda_case(7 * D + Lq, B=B, Lq=Lq, H=H, D=D)
    loc[:, Lq // 2:, :, :, :] = 1e-5                         # second half of every image: everything on the corner pixel
    loc[:, 3:7] = loc[:, 3:4]                                # four identical neighbours somewhere else
    q1 = min(8, Lq - 1)
    loc[:, q1, :, :, :] = loc[:, q1, :, :1, :]               # one query whose points all coincide
    if B > 1:
        loc[1, :2] = 1e-5                                    # the run of image 0 continues into image 1
    if dtype == torch.bfloat16:
        value = torch.tensor(value).bfloat16().float().numpy()
        go = torch.tensor(go).bfloat16().float().numpy()
    out, gv, gl, gw = _run(value, loc, w, go, shapes, points, cuda, dtype)
    rv, rl, rw = np_ref.msda_backward(value, shapes, loc, w, points, go)
    if dtype == torch.float32:
        np.testing.assert_allclose(gv, rv, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(rv).max()))
        np.testing.assert_allclose(gw, rw, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gl, rl, rtol=1e-4, atol=3e-4)
    else:       # f16 accumulator + bf16 result: hot rows collect B-independent sums of ~Lq / 2 * P addends
        assert np.abs(gv - rv).max() <= 1e-2 * np.abs(rv).max()
        assert np.abs(gw - rw).max() <= 2e-2 * max(1.0, np.abs(rw).max())
