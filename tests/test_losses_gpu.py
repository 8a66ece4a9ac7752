"""A13/A14 parity: fused HIP head-loss kernels (through DFINECriterion on cuda tensors) vs the
golden loss dicts / gradients of the reference criterion and vs this build's torch composition."""
import numpy as np
import pytest
import torch

from custom_d_fine_amd.d_fine import dfine
from tests import helpers

pytestmark = pytest.mark.gpu
G = helpers.GOLDEN_DIR


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_criterion_matches_reference_golden(cuda, seed, dtype):
    g = np.load(f"{G}/criterion.npz")
    crit = dfine.build_loss("s", 6, 0.0, False)
    outputs = helpers.make_criterion_outputs(seed, device=cuda)
    if dtype == torch.bfloat16:     # storage in bf16 like an autocast forward; boxes / refs stay fp32
        def cast(d):
            for k in ("pred_logits", "pred_corners"):
                if k in d:
                    d[k] = d[k].detach().to(dtype).requires_grad_(True)
        cast(outputs)
        for lst in ("aux_outputs", "dn_outputs", "enc_aux_outputs"):
            for d in outputs[lst]:
                cast(d)
        cast(outputs["pre_outputs"]); cast(outputs["dn_pre_outputs"])
        for d in outputs["aux_outputs"]:
            d["teacher_corners"], d["teacher_logits"] = outputs["pred_corners"], outputs["pred_logits"]
        for d in outputs["dn_outputs"]:
            d["teacher_corners"], d["teacher_logits"] = outputs["dn_outputs"][-1]["pred_corners"], outputs["dn_outputs"][-1]["pred_logits"]
    targets, meta = helpers.criterion_targets_and_meta(device=cuda)
    outputs["dn_meta"] = meta
    assert crit._fusable(outputs)
    losses = crit(outputs, targets)
    want = {k.split("/", 2)[2]: float(g[k]) for k in g.files if k.startswith(f"s{seed}/loss/")}
    assert set(losses) == set(want)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for k, v in want.items():
        assert abs(losses[k].item() - v) < tol * max(1.0, abs(v)), (k, losses[k].item(), v)
    crit.total(losses).backward()
    got = {"pred_logits": outputs["pred_logits"].grad, "pred_boxes": outputs["pred_boxes"].grad,
           "pred_corners": outputs["pred_corners"].grad,
           "aux0_corners": outputs["aux_outputs"][0]["pred_corners"].grad,
           "dn0_logits": outputs["dn_outputs"][0]["pred_logits"].grad,
           "enc_boxes": outputs["enc_aux_outputs"][0]["pred_boxes"].grad}
    for k, v in got.items():
        ref = g[f"s{seed}/grad/{k}"]
        if dtype == torch.float32:
            np.testing.assert_allclose(v.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-5)
        else:
            cos = torch.nn.functional.cosine_similarity(v.float().cpu().flatten(), torch.tensor(ref).flatten(), dim=0)
            assert cos > 0.999, (k, cos)


def test_fused_equals_torch_composition_on_strided_views(cuda):
    """The decoder hands the criterion strided views (dn / matching split of stacked tensors)."""
    torch.manual_seed(0)
    crit = dfine.build_loss("s", 6, 0.0, False)
    out = helpers.make_criterion_outputs(3, device=cuda, requires_grad=False)
    # re-create the main head as the second half of a wider tensor -> non-contiguous views
    wide_logits = torch.randn(2, 40, 6, device=cuda, requires_grad=True)
    wide_corners = torch.randn(2, 40, 132, device=cuda, requires_grad=True)
    out["pred_logits"], out["pred_corners"] = wide_logits[:, 16:], wide_corners[:, 16:]
    for d in out["aux_outputs"]:
        d["teacher_corners"], d["teacher_logits"] = out["pred_corners"], out["pred_logits"]
    targets, meta = helpers.criterion_targets_and_meta(device=cuda)
    out["dn_meta"] = meta
    fused = crit(out, targets)
    crit.total(fused).backward()
    g_fused = (wide_logits.grad.clone(), wide_corners.grad.clone())
    wide_logits.grad = wide_corners.grad = None
    crit._fusable = lambda o: False
    plain = crit(out, targets)
    sum(plain.values()).backward()
    for k in plain:
        assert abs(plain[k].item() - fused[k].item()) < 1e-4 * max(1.0, abs(plain[k].item())), k
    assert torch.allclose(g_fused[0], wide_logits.grad, rtol=1e-3, atol=1e-6)
    assert torch.allclose(g_fused[1], wide_corners.grad, rtol=1e-3, atol=1e-6)
    assert (g_fused[0][:, :16] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_ddf,with_corners", [(True, True), (False, True), (False, False)])
def test_head_grads_scale_matches_torch_composition(cuda, dtype, with_ddf, with_corners):
    """dfine_head_grads_scale (the one-launch backward of the fused head losses) against the element-wise composition it
    replaces; sizes that are not multiples of the 4-element vectors."""
    from custom_d_fine_amd import hip
    torch.manual_seed(3)
    g = torch.randn(5, device=cuda)
    nl, nb, nc = 2 * 37 * 7 + 1, 2 * 37 * 4, 2 * 37 * 132 + 3
    gl = torch.randn(nl, device=cuda).to(dtype)
    l1, gi = torch.randn(nb, device=cuda), torch.randn(nb, device=cuda)
    gf = torch.randn(nc, device=cuda).to(dtype) if with_corners else None
    gd = torch.randn(nc, device=cuda).to(dtype) if with_ddf else None
    want_l = gl.float() * g[0]
    want_b = l1 * g[1] + gi * g[2]
    want_c = None if gf is None else gf.float() * g[3] + (gd.float() * g[4] if gd is not None else 0)
    hip.head_grads_scale(g, gl, l1, gi, gf, gd)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (gl.float() - want_l).abs().max() <= tol * want_l.abs().max()
    assert (l1 - want_b).abs().max() <= 1e-6 * want_b.abs().max()
    if gf is not None:
        assert (gf.float() - want_c).abs().max() <= tol * want_c.abs().max()
