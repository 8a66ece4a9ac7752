"""A10 / A15 kernels of the segmentation head (csrc/mask.hip, the per-image-weight 1x1 convolution of conv.hip) through the
C ABI against plain PyTorch fp32 references of the same ops (the reference's own composition: nn.GroupNorm, F.interpolate,
torch.einsum, binary_cross_entropy_with_logits, the matcher's dice / focal cost functions restated in the test)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from custom_d_fine_amd import kernels

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return F.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 12, 20), (3, 256, 30, 30), (1, 32, 7, 9)])
def test_groupnorm_forward_backward(cuda, dtype, relu, shape):
    torch.manual_seed(0)
    gn = nn.GroupNorm(32, shape[1]).to(cuda)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_(0, 0.3)
    x0 = (torch.randn(shape, device=cuda) * 2 + 0.7)
    x = x0.to(dtype).requires_grad_(True)
    go = torch.randn(shape, device=cuda)
    y = kernels.group_norm_act(x, gn, relu)
    assert y.dtype == dtype
    y.backward(go.to(dtype))
    got = (y.detach().float(), x.grad.float(), gn.weight.grad.clone(), gn.bias.grad.clone())
    gn.zero_grad()
    xr = x.detach().float().requires_grad_(True)                 # reference on the values the kernel saw
    yr = gn(xr)
    if relu:
        yr = F.relu(yr)
    yr.backward(go.to(dtype).float())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (got[0] - yr.detach()).abs().max() <= tol * max(yr.abs().max().item(), 1.0)
    assert (got[1] - xr.grad).abs().max() <= (1e-4 if dtype == torch.float32 else 3e-2) * max(xr.grad.abs().max().item(), 1e-6)
    np.testing.assert_allclose(got[2].cpu().numpy(), gn.weight.grad.cpu().numpy(), rtol=2e-3 if dtype == torch.float32 else 3e-2,
                               atol=(1e-3 if dtype == torch.float32 else 5e-2) * gn.weight.grad.abs().max().item())
    np.testing.assert_allclose(got[3].cpu().numpy(), gn.bias.grad.cpu().numpy(), rtol=2e-3 if dtype == torch.float32 else 3e-2,
                               atol=(1e-3 if dtype == torch.float32 else 5e-2) * gn.bias.grad.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hin,hout", [((15, 15), (30, 30)), ((30, 30), (120, 120)), ((8, 12), (32, 48)), ((7, 5), (20, 13)),
                                      ((40, 40), (10, 10)), ((9, 9), (9, 9)), ((1, 1), (4, 4)), ((60, 60), (120, 120))])
def test_bilinear_resize_forward_backward(cuda, dtype, hin, hout):
    torch.manual_seed(1)
    x = torch.randn(2, 5, *hin, device=cuda).to(dtype).requires_grad_(True)
    base = torch.randn(2, 5, *hout, device=cuda).to(dtype).requires_grad_(True)
    go = torch.randn(2, 5, *hout, device=cuda).to(dtype)
    y = kernels.bilinear_resize(x, hout, base=base)
    y.backward(go)
    xr = x.detach().float().requires_grad_(True)
    yr = base.detach().float() + F.interpolate(xr, size=hout, mode="bilinear", align_corners=False)
    yr.backward(go.float())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (y.detach().float() - yr.detach()).abs().max() <= tol * max(yr.abs().max().item(), 1.0)
    assert (x.grad.float() - xr.grad).abs().max() <= (1e-5 if dtype == torch.float32 else 2e-2) * max(xr.grad.abs().max().item(), 1.0)
    assert torch.equal(base.grad, go)
    y2 = kernels.bilinear_resize(x.detach(), hout)               # without a base
    assert (y2.float() - (yr.detach() - base.detach().float())).abs().max() <= 2 * tol * max(yr.abs().max().item(), 1.0)


@pytest.mark.parametrize("B,Q,C,H,W", [(2, 300, 256, 40, 40), (3, 52, 64, 24, 32), (2, 196, 256, 80, 80)])
def test_mask_logits_einsum(cuda, B, Q, C, H, W):
    torch.manual_seed(2)
    emb = (torch.randn(B, Q, C, device=cuda) / C ** 0.5).requires_grad_(True)
    feat = torch.randn(B, C, H, W, device=cuda).bfloat16().requires_grad_(True)
    go = torch.randn(B, Q, H, W, device=cuda).bfloat16()
    y = kernels.mask_logits(emb, feat)
    assert y.dtype == torch.bfloat16 and y.shape == (B, Q, H, W)
    y.backward(go)
    e32 = emb.detach().bfloat16().float().requires_grad_(True)   # the operands the MFMA kernel multiplies
    f32 = feat.detach().float().requires_grad_(True)
    yr = torch.einsum("bqc,bchw->bqhw", e32, f32)
    yr.backward(go.float())
    assert (y.detach().float() - yr.detach()).abs().max() <= 2e-2 * yr.abs().max()
    assert _cos(feat.grad, f32.grad) > 0.9999 and (feat.grad.float() - f32.grad).abs().max() <= 3e-2 * f32.grad.abs().max()
    assert _cos(emb.grad, e32.grad) > 0.9999 and (emb.grad - e32.grad).abs().max() <= 2e-2 * e32.grad.abs().max()


def test_wide_3x3_convolution_halves(cuda):
    """The 240-pixel-wide maps of the head at 960 x 960: two overlapping column halves through the 160-pixel strip kernels."""
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 48, 3, padding=1, bias=False).to(cuda)
    x = torch.randn(2, 64, 36, 240, device=cuda).bfloat16().requires_grad_(True)
    go = torch.randn(2, 48, 36, 240, device=cuda).bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.conv_plain(x, conv)
    assert y.shape == (2, 48, 36, 240)
    y.backward(go)
    got_w = conv.weight.grad.clone()
    conv.zero_grad()
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, conv.weight.detach().bfloat16().float(), None, 1, 1)
    wr = conv.weight.detach().bfloat16().float().requires_grad_(True)
    F.conv2d(xr.detach(), wr, None, 1, 1).backward(go.float())
    yr.backward(go.float())
    assert (y.detach().float() - yr.detach()).abs().max() <= 2e-2 * yr.abs().max()
    assert (x.grad.float() - xr.grad).abs().max() <= 2e-2 * xr.grad.abs().max()
    assert _cos(got_w, wr.grad) > 0.9999 and (got_w - wr.grad).abs().max() <= 1e-2 * wr.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_losses_match_torch_composition(cuda, dtype):
    from custom_d_fine_amd.d_fine.dfine_criterion import DFINECriterion
    torch.manual_seed(4)
    B, Q, H, W, M = 3, 20, 48, 40, 9
    pm = (torch.randn(B, Q, H, W, device=cuda) * 2).to(dtype).requires_grad_(True)
    pb = torch.randint(0, B, (M,), device=cuda)
    pq = torch.randperm(Q, device=cuda)[:M]
    tgt = (torch.rand(M, H, W, device=cuda) > 0.6).float() * torch.rand(M, H, W, device=cuda).clamp(min=0.3)
    x1, y1 = torch.rand(M, device=cuda) * 20, torch.rand(M, device=cuda) * 20
    boxes = torch.stack([x1, y1, x1 + 2 + torch.rand(M, device=cuda) * 18, y1 + 2 + torch.rand(M, device=cuda) * 25], 1)
    boxes[0] = torch.tensor([3.0, 5.0, 3.4, 5.2])                 # covers no pixel centre: area clamps to 1, sums are 0
    bce, dice = kernels.mask_losses(pm, pb, pq, tgt, boxes)
    (1.7 * bce + 0.6 * dice).backward()
    # the same through a target-row indirection (targets of the whole batch prepared once, rows picked by the plan)
    perm = torch.randperm(M, device=cuda)
    inv = torch.argsort(perm)
    pm2 = pm.detach().clone().requires_grad_(True)
    bce2, dice2 = kernels.mask_losses(pm2, pb, pq, tgt[perm], boxes[perm], plan_t=inv)
    (1.7 * bce2 + 0.6 * dice2).backward()
    assert torch.equal(bce2, bce) and torch.equal(dice2, dice) and torch.equal(pm2.grad, pm.grad)
    pr = pm.detach().float().requires_grad_(True)
    sel = pr[pb, pq]
    rb, rd = DFINECriterion._cropped_bce_loss(sel, tgt, boxes), DFINECriterion._cropped_dice_loss(sel, tgt, boxes)
    (1.7 * rb + 0.6 * rd).backward()
    assert abs(bce.item() - rb.item()) <= 1e-5 * max(abs(rb.item()), 1) and abs(dice.item() - rd.item()) <= 1e-5
    tol = 1e-6 if dtype == torch.float32 else 1e-2 * pr.grad.abs().max().item()
    assert (pm.grad.float() - pr.grad).abs().max() <= max(tol, 1e-6)
    assert (pm.grad.float() != 0).any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_cost_sums_match_matcher_functions(cuda, dtype):
    from custom_d_fine_amd.d_fine.matcher import dice_cost, sigmoid_focal_cost
    torch.manual_seed(5)
    B, Qall, Q, H, W = 3, 30, 22, 30, 34
    sizes = [5, 0, 37]
    pm = (torch.randn(B, Qall, H, W, device=cuda) * 3).to(dtype)
    gt = (torch.rand(sum(sizes), H, W, device=cuda) > 0.7).float()
    toff = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), device=cuda)
    out, qsum = kernels.mask_cost_sums(pm, gt, toff, Q, max(sizes), 0.25, 2.0)
    off = 0
    for b, n in enumerate(sizes):
        if n == 0:
            continue
        x = pm[b, Qall - Q:].float()
        g = gt[off: off + n]
        off += n
        want_d = dice_cost(x.sigmoid(), g)
        got_d = 1 - (2 * out[b, :, :n, 0] + 1e-6) / (qsum[b, :, None, 0] + g.flatten(1).sum(1)[None, :] + 1e-6)
        want_f = sigmoid_focal_cost(x.flatten(1), g.flatten(1), 0.25, 2.0)
        got_f = (out[b, :, :n, 1] + qsum[b, :, None, 1]) / (H * W)
        np.testing.assert_allclose(got_d.cpu().numpy(), want_d.cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(got_f.cpu().numpy(), want_f.cpu().numpy(), rtol=2e-4, atol=2e-6)
