"""fp32 token-stream GEMMs (csrc/gemm_f32.hip, BASELINE config #2) through the C ABI against fp64 torch: the NT GEMM with
bias / activation / batch / split partial products, the nn.Linear autograd node and the attention composition built on it,
and - model level - that an fp32 train step of D-FINE-s launches no rocBLAS / hipBLASLt / library attention kernel."""
import pytest
import torch
import torch.nn.functional as F

from custom_d_fine_amd import hip, kernels

pytestmark = pytest.mark.gpu


def _close(got, want, tol=2e-6):
    want = want.to(torch.float64)
    scale = want.abs().max().clamp_min(1e-12)
    assert ((got.to(torch.float64) - want).abs().max() / scale).item() < tol


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (130, 70, 33), (15744 // 8, 256, 256), (5, 3, 7), (1, 132, 256), (300, 80, 4)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_nt_bias_act(cuda, M, N, K, act):
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=cuda)
    b = torch.randn(N, K, device=cuda)
    bias = torch.randn(N, device=cuda)
    got = hip.gemm_f32_nt(a, b, bias, alpha=0.5, act=act)
    pre = 0.5 * (a.double() @ b.double().t()) + bias.double()
    want = {0: lambda t: t, 1: F.relu, 2: F.gelu, 3: F.silu}[act](pre)
    _close(got, want, 3e-6)


def test_gemm_nt_strided_rows_batch_and_splits(cuda):
    torch.manual_seed(3)
    big = torch.randn(200, 96, device=cuda)
    a = big[:, 16:80]                                   # row stride 96, K = 64, base not 16-byte aligned rows? (64-byte offset)
    b = torch.randn(40, 64, device=cuda)
    _close(hip.gemm_f32_nt(a, b), a.double() @ b.double().t())
    a3 = torch.randn(6, 50, 32, device=cuda)
    b3 = torch.randn(6, 70, 32, device=cuda)
    _close(hip.gemm_f32_nt(a3, b3, alpha=2.0), 2.0 * torch.einsum("zmk,znk->zmn", a3.double(), b3.double()))
    _close(hip.gemm_f32_nt(a3, b3[0]), torch.einsum("zmk,nk->zmn", a3.double(), b3[0].double()))       # shared second operand
    x = torch.randn(48, 5000, device=cuda)
    y = torch.randn(72, 5000, device=cuda)
    part = hip.gemm_f32_nt(x, y, splits=13)
    assert part.dim() == 3 and part.shape[1:] == (48, 72)
    _close(part.sum(0), x.double() @ y.double().t(), 5e-6)


@pytest.mark.parametrize("act", [None, "relu", "gelu", "silu"])
def test_linear_fp32_autograd(cuda, act):
    torch.manual_seed(4)
    lin = torch.nn.Linear(96, 72).to(cuda)
    x = torch.randn(3, 50, 96, device=cuda, requires_grad=True)
    go = torch.randn(3, 50, 72, device=cuda)
    y = kernels.linear(x, lin.weight, lin.bias, act)
    y.backward(go)
    got = (y.detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    xr = x.detach().double().requires_grad_(True)
    wr, br = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    if act:
        yr = getattr(F, act)(yr)
    yr.backward(go.double())
    for g, w in zip(got, (yr.detach(), xr.grad, wr.grad, br.grad)):
        _close(g, w, 5e-6)


@pytest.mark.parametrize("masked", [False, True])
def test_attention_fp32_composition(cuda, masked):
    torch.manual_seed(5)
    B, H, L, d = 2, 8, 77, 32
    q, k, v = (torch.randn(B, H, L, d, device=cuda, requires_grad=True) for _ in range(3))
    allowed = None
    if masked:
        allowed = torch.rand(L, L, device=cuda) > 0.3
        allowed |= torch.eye(L, device=cuda, dtype=torch.bool)
    go = torch.randn(B, H, L, d, device=cuda)
    o = kernels.attention_f32(q, k, v, allowed)
    o.backward(go)
    got = (o.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone())
    qr, kr, vr = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    s = qr @ kr.transpose(-1, -2) / d ** 0.5
    if masked:
        s = s.masked_fill(~allowed, float("-inf"))
    orf = torch.softmax(s, -1) @ vr
    orf.backward(go.double())
    for g, w in zip(got, (orf.detach(), qr.grad, kr.grad, vr.grad)):
        _close(g, w, 1e-5)


def test_fp32_train_step_launches_no_library_gemm(cuda):
    """D-FINE-s fp32 (BASELINE configs[1]) 320 x 320: one profiled train step - no rocBLAS / hipBLASLt (Tensile `Cijk_`) kernel,
    no MIOpen convolution, no library attention kernel."""
    import bench
    from custom_d_fine_amd.dl.synthetic import make_batch
    step = bench.build_step("s", 320, cuda, None)
    images, targets = make_batch(4, 320, seed=7, device=cuda)
    step(images, targets)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        loss, _ = step(images, targets)
        torch.cuda.synchronize()
    assert torch.isfinite(loss)
    names = [e.key for e in prof.key_averages()]
    bad = [n for n in names if any(p in n for p in ("Cijk_", "rocblas", "gemv", "miopen", "Miopen", "igemm", "aotriton", "attn_fwd", "flash"))
           and "dfine::" not in n]
    assert not bad, bad
    assert any("gemm_f32_nt_kernel" in n for n in names) and any("conv_f32_kernel" in n for n in names)


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("batch", [1, 5])
def test_gemm_f32_operand_layouts(cuda, akm, bkm, batch):
    """K-major operands ([K, M] / [K, N]) read in place: every transposition combination against fp64 matmul."""
    torch.manual_seed(11)
    M, N, K = 70, 45, 130
    a = torch.randn(*((batch,) if batch > 1 else ()), *((K, M) if akm else (M, K)), device=cuda)
    b = torch.randn(*((batch,) if batch > 1 else ()), *((K, N) if bkm else (N, K)), device=cuda)
    got = hip.gemm_f32(a, b, a_kmajor=akm, b_kmajor=bkm, alpha=1.5)
    ad = a.double().transpose(-1, -2) if akm else a.double()
    bd = b.double() if bkm else b.double().transpose(-1, -2)
    _close(got, 1.5 * (ad @ bd), 3e-6)
    if batch == 1:
        part = hip.gemm_f32(a, b, a_kmajor=akm, b_kmajor=bkm, splits=4)
        _close(part.sum(0), ad @ bd, 3e-6)


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,batch", [(130, 200, 37, 30), (256, 384, 64, 4), (97, 1601, 130, 2), (512, 400, 515, 16)])
def test_gemm_f32_large_tile_kernel(cuda, akm, bkm, M, N, K, batch):
    """Shapes that take the 128 x 128-tile kernel (gemm_f32_big_kernel: M, N >= 96 and >= 224 workgroups): edge tiles in both
    directions, K tails that are not multiples of 4 / 16, every operand layout, batches, against the fp64 product."""
    torch.manual_seed(M + N + K)
    a = torch.randn(batch, *((K, M) if akm else (M, K)), device=cuda)
    b = torch.randn(batch, *((K, N) if bkm else (N, K)), device=cuda)
    bias = torch.randn(N, device=cuda)
    got = hip.gemm_f32(a, b, a_kmajor=akm, b_kmajor=bkm, bias=bias, alpha=0.75, act=1)
    ad = a.double().transpose(-1, -2) if akm else a.double()
    bd = b.double() if bkm else b.double().transpose(-1, -2)
    _close(got, torch.relu(0.75 * (ad @ bd) + bias.double()), 3e-6)


def test_gemm_f32_large_tile_splits_and_shared_operand(cuda):
    """K-split partial products (weight-gradient form) and a 2-D second operand shared by the batch on the large-tile kernel."""
    torch.manual_seed(21)
    dy, x = torch.randn(16, 256, 1600, device=cuda), torch.randn(16, 384, 1600, device=cuda)
    part = hip.gemm_f32_nt(dy, x, splits=3)                               # [16 * 3, 256, 384]
    assert part.shape == (48, 256, 384)
    _close(part.view(16, 3, 256, 384).sum(1), dy.double() @ x.double().transpose(1, 2), 3e-6)
    w = torch.randn(640, 256, device=cuda)
    y = hip.conv1x1_f32(dy.view(16, 256, 40, 40), w)                     # shared row-major A, K-major B per image
    _close(y.view(16, 640, 1600), w.double() @ dy.double(), 3e-6)
    tok = torch.randn(134400 // 8, 256, device=cuda)
    _close(hip.gemm_f32_nt(tok, w), tok.double() @ w.double().t(), 3e-6)


@pytest.mark.parametrize("M,N,relu", [(7968, 256, False), (7968, 1024, True), (333, 80, True), (31, 4, False), (4800, 132, False)])
def test_colsum_f32_bias_gradient_partials(cuda, M, N, relu):
    """dfine_colsum_f32: per-split column sums (+ the ReLU mask of the layer's output) against fp64 torch."""
    torch.manual_seed(M + N)
    d = torch.randn(M, N, device=cuda)
    y = torch.randn(M, N, device=cuda).relu() if relu else None
    assert hip.colsum_f32_ok(d)
    part, dm = hip.colsum_f32(d, y)
    want_dm = d * (y > 0) if relu else d
    assert torch.equal(dm, want_dm)
    want = want_dm.double().sum(0)
    assert (part.double().sum(0) - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0) * (M ** 0.5)
