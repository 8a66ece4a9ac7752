"""Fused residual / gate + LayerNorm kernels (csrc/lnfused.hip) against the torch composition they replace
(src/d_fine/arch/dfine_decoder.py:238-271), forward and backward, fp32 / bf16 operand mixes."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    return (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item())


@pytest.mark.parametrize("D", [128, 256, 384])
@pytest.mark.parametrize("branch_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("clamp", [None, 3.0])
def test_add_layer_norm(cuda, D, branch_dtype, clamp):
    from custom_d_fine_amd import kernels
    torch.manual_seed(D)
    norm = nn.LayerNorm(D).to(cuda)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(3, 37, D, device=cuda) * 2).requires_grad_(True)
    r = (torch.randn(3, 37, D, device=cuda) * 2).to(branch_dtype).requires_grad_(True)
    y = kernels.add_layer_norm(x, r, norm, clamp=clamp)
    xr, rr = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
    ref = nn.LayerNorm(D).to(cuda)
    ref.load_state_dict(norm.state_dict())
    z = xr + rr
    if clamp is not None:
        z = z.clamp(min=-clamp, max=clamp)
    yr = ref(z)
    assert y.dtype == torch.float32 and _close(y, yr, 2e-5)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert _close(x.grad, xr.grad, 2e-5)
    assert r.grad.dtype == branch_dtype and _close(r.grad, rr.grad, 2e-5 if branch_dtype == torch.float32 else 1e-2)
    assert _close(norm.weight.grad, ref.weight.grad, 1e-4) and _close(norm.bias.grad, ref.bias.grad, 1e-4)


@pytest.mark.parametrize("g_dtype,x2_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_gate_layer_norm(cuda, g_dtype, x2_dtype):
    from custom_d_fine_amd import kernels
    torch.manual_seed(3)
    D = 256
    norm = nn.LayerNorm(D).to(cuda)
    x1 = torch.randn(2, 50, D, device=cuda).requires_grad_(True)
    x2 = torch.randn(2, 50, D, device=cuda).to(x2_dtype).requires_grad_(True)
    g = (torch.randn(2, 50, 2 * D, device=cuda) * 2).to(g_dtype).requires_grad_(True)
    y = kernels.gate_layer_norm(g, x1, x2, norm)
    x1r, x2r, gr = (t.detach().clone().requires_grad_(True) for t in (x1, x2, g))
    ref = nn.LayerNorm(D).to(cuda)
    ref.load_state_dict(norm.state_dict())
    s1, s2 = torch.sigmoid(gr.float()).chunk(2, dim=-1)
    yr = ref(s1 * x1r + s2 * x2r.float())
    assert _close(y, yr, 3e-5)
    go = torch.randn_like(yr)
    y.backward(go)
    yr.backward(go)
    tol = 3e-5 if g_dtype == torch.float32 else 1e-2
    assert _close(x1.grad, x1r.grad, 3e-5) and _close(x2.grad, x2r.grad, tol) and _close(g.grad, gr.grad, tol)
    assert _close(norm.weight.grad, ref.weight.grad, 1e-4) and _close(norm.bias.grad, ref.bias.grad, 1e-4)
