"""A5/A6 parity: split-K weight/bias-gradient kernel of the token-stream linears vs a plain PyTorch fp32
reference on the bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from custom_d_fine_amd import hip as hipmod
from custom_d_fine_amd import kernels

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(15744, 256, 256), (15744, 1024, 256), (4100, 132, 256), (5000, 512, 4),
                                   (4099, 64, 20), (8192, 1, 64), (12800, 80, 256)])
def test_linear_wgrad_kernel(cuda, M, N, K):
    torch.manual_seed(N + K)
    x = torch.randn(M, K, device=cuda).bfloat16()
    dy = torch.randn(M, N, device=cuda).bfloat16()
    dw, db = hipmod.linear_wgrad_bf16(x, dy, with_bias=True)
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    assert torch.allclose(dw, ref_w, rtol=1e-3, atol=1e-3 * ref_w.abs().max().item())
    assert torch.allclose(db, ref_b, rtol=1e-3, atol=1e-3 * ref_b.abs().max().item())


def test_linear_function_matches_f_linear(cuda):
    """kernels.linear under bf16 autocast: a fresh (non-view) bf16 output that callers may modify in place (the MLP heads'
    ReLU(inplace=True)), and gradients equal to F.linear's on the same bf16-rounded operands.  (No ReLU in the numeric
    comparison: this kernel adds the fp32 bias before the one rounding to bf16, hipBLASLt rounds the bias to bf16 first, so
    outputs within one bf16 ulp of zero land on different sides and flip their ReLU mask.)"""
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 132).to(cuda)
    x = torch.randn(16, 300, 256, device=cuda, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.linear(x, lin.weight, lin.bias)
        assert y.dtype == torch.bfloat16 and y._base is None
    go = torch.randn_like(y)
    y.backward(go)
    g = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        F.relu(kernels.linear(x, lin.weight, lin.bias), inplace=True).sum().backward()      # in-place on the output is legal
    x.grad = None; lin.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        F.linear(x, lin.weight, lin.bias).backward(go)
    for a, b in zip(g, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * b.abs().max().item())


def test_clamp_pos_matches_aten_clamp(cuda):
    """kernels.clamp_pos (the decoder's clamp of the query position embedding, one launch each way) against x.clamp(-10, 10)
    and ATen's ClampBackward1: values beyond, inside and exactly on the bounds."""
    from custom_d_fine_amd import kernels
    torch.manual_seed(0)
    x = (torch.randn(4, 123, 256, device=cuda) * 8).bfloat16()
    x.view(-1)[:6] = torch.tensor([10.0, -10.0, 10.0625, -10.0625, 9.9375, -9.9375], device=cuda).bfloat16()
    g = torch.randn(4, 123, 256, device=cuda).bfloat16()
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya, yb = kernels.clamp_pos(a), b.clamp(min=-10, max=10)
    assert torch.equal(ya, yb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(a.grad, b.grad)
    assert (a.grad.view(-1)[:2] == g.view(-1)[:2]).all() and (a.grad.view(-1)[2:4] == 0).all()


@pytest.mark.parametrize("n,rows,D,pad", [(6272, 81, 256, -1), (1000, 5, 100, 2), (3, 81, 256, -1), (0, 7, 64, -1), (513, 300, 300, 0)])
def test_embedding_backward_scan_kernel(cuda, n, rows, D, pad):
    """dfine_embedding_bwd (one scan kernel per small table, lookup order, no atomics) against ATen's embedding_dense_backward
    and against the autograd path of kernels.embedding (the denoising class embedding of the decoder)."""
    from custom_d_fine_amd import hip, kernels
    torch.manual_seed(n + rows)
    idx = torch.randint(0, rows, (n,), device=cuda)
    g = torch.randn(n, D, device=cuda)
    got = hip.embedding_backward(g, idx, rows, pad)
    assert torch.equal(got, hip.embedding_backward(g, idx.int(), rows, pad))          # 32-bit indices: the same kernel, the same order
    want = torch.ops.aten.embedding_dense_backward(g, idx, rows, pad, False) if n else torch.zeros(rows, D, device=cuda)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5 * max(1.0, float(want.abs().max())))
    if n and pad < 0:
        emb = torch.nn.Embedding(rows, D).to(cuda)
        out = kernels.embedding(emb, idx.view(1, -1))
        out.backward(g.view(1, n, D))
        torch.cuda.synchronize()
        assert torch.allclose(emb.weight.grad, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("dims,rows", [((256, 1024, 256), (32, 496)), ((256, 256, 256, 132), (32, 496)), ((4, 512, 256), (32, 496)),
                                       ((20, 64, 1), (7, 301)), ((256, 256, 4), (3, 300))])
def test_mlp_relu_fused_backward_matches_per_layer_composition(cuda, dims, rows):
    """kernels.mlp_relu (one autograd node, the ReLU backward in the data-gradient epilogue: dfine_linear_dgrad_relu) against the
    per-layer kernels.linear composition it replaces: outputs and every gradient BIT-identical (the mask is applied before the one
    rounding to bf16), for the MLP shapes of the decoder (FFN, box heads, query position head 4 -> 512 -> 256, LQE 20 -> 64 -> 1);
    and the masked data gradient against fp32 math on the same bf16-rounded operands."""
    torch.manual_seed(len(dims) + dims[0])
    layers = [torch.nn.Linear(a, b).to(cuda) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(*rows, dims[0], device=cuda)

    def run(fused):
        xin = x.clone().requires_grad_(True)
        for l in layers:
            l.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                y = kernels.mlp_relu(xin, layers)
            else:
                h = xin
                for l in layers[:-1]:
                    h = kernels.linear(h, l.weight, l.bias, act="relu")
                y = kernels.linear(h, layers[-1].weight, layers[-1].bias)
        go = torch.randn(y.shape, device=cuda, generator=torch.Generator(device="cuda").manual_seed(5)).to(y.dtype)
        y.backward(go)
        return [y.detach(), xin.grad] + [l.weight.grad.clone() for l in layers] + [l.bias.grad.clone() for l in layers]

    a, b = run(True), run(False)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and torch.equal(u, v), f"tensor {i}: fused MLP differs from the per-layer composition"
    # the epilogue form on its own, vs fp32
    M, K, N = 4099, 132, 256
    d2 = torch.randn(M, K, device=cuda).bfloat16()
    wt = torch.randn(N, K, device=cuda).bfloat16()
    h = torch.randn(M, N, device=cuda).relu().bfloat16()
    h[0, :8] = torch.tensor([0.0, -0.0, 1e-30, float("inf"), float("nan"), 1.0, -1.0, 0.5], device=cuda).bfloat16()
    got = hipmod.linear_dgrad_relu(d2, wt, h).float()
    ref = (d2.float() @ wt.float().t()) * (h.float() > 0)
    assert torch.allclose(got, ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item())
    assert torch.equal(got == 0, ~(h.float() > 0) | (ref == 0))
