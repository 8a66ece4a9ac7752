"""End-to-end parity on the GPU: D-FINE forward / train step through the HIP kernels vs the golden
vectors generated from the reference (fp32, north_star tolerance 1e-3 on logits/boxes)."""
import os

import numpy as np
import pytest
import torch

from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.d_fine.arch import utils as U
from tests import helpers
from tests.test_model_cpu import assert_same_query_set

pytestmark = pytest.mark.gpu
G = helpers.GOLDEN_DIR


@pytest.fixture(autouse=True)
def _fp32_math():
    old = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    yield
    torch.backends.cuda.matmul.allow_tf32 = old


def test_native_library_is_loaded():
    import custom_d_fine_amd.hip as h
    assert h._lib.dfine_abi_version() == h.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libdfine_hip.so" in maps


@pytest.mark.parametrize("size,img,batch,name", [("n", 320, 2, "model_n320.npz"), ("m", 640, 1, "model_m640_eval.npz"),
                                                 ("m", 640, 3, "model_m640_eval_b3.npz")])
def test_eval_forward_matches_reference(cuda, size, img, batch, name):
    """fp32 eval forward of the headline model at full input size against the reference's outputs (1 and 3 images), every
    query within 1e-3 (north_star) - boundary swaps of the top-300 selection aside, which assert_same_query_set tells from
    real tolerance violations."""
    g = np.load(f"{G}/{name}")
    m = dfine.build_model(size, 80, False, "cpu", img_size=[img, img])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).eval()
    with torch.no_grad():
        o = m(helpers.make_images(batch, img).to(cuda))
    assert_same_query_set(o["pred_logits"].cpu(), o["pred_boxes"].cpu(), torch.tensor(g["eval/pred_logits"]),
                          torch.tensor(g["eval/pred_boxes"]))


def test_train_step_matches_reference_n320(cuda):
    g = np.load(f"{G}/model_n320.npz")
    m = dfine.build_model("n", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    crit = dfine.build_loss("n", 80, 0.0, False)
    targets = helpers.make_targets(2, 80, device=cuda)
    U.set_denoising_generator(torch.Generator().manual_seed(11))   # the reference's CPU noise stream
    try:
        out = m(helpers.make_images(2, 320).to(cuda), targets)
    finally:
        U.set_denoising_generator(None)
    losses = crit(out, targets)
    want = {k.split("/", 2)[2]: float(g[k]) for k in g.files if k.startswith("train/loss/")}
    assert set(losses) == set(want)
    for k, v in want.items():
        assert abs(losses[k].item() - v) < 2e-3 * max(1.0, abs(v)), (k, losses[k].item(), v)
    sum(losses.values()).backward()
    params = dict(m.named_parameters())
    for k in [f for f in g.files if f.startswith("train/grad/")]:
        name = k.split("/", 2)[2]
        ref = torch.tensor(g[k])
        got = params[name].grad.cpu()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        assert cos > 0.9999, (name, cos)


def _train_step_vs_golden(cuda, size, golden, amp, loss_tol, cos_min):
    g = np.load(f"{G}/{golden}")
    m = dfine.build_model(size, 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    crit = dfine.build_loss(size, 80, 0.0, False)
    targets = helpers.make_targets(2, 80, device=cuda)
    U.set_denoising_generator(torch.Generator().manual_seed(11))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = m(helpers.make_images(2, 320).to(cuda), targets)
    finally:
        U.set_denoising_generator(None)
    losses = crit(out, targets)
    want = {k.split("/", 2)[2]: float(g[k]) for k in g.files if k.startswith("train/loss/")}
    assert set(losses) == set(want)
    for k, v in want.items():
        assert abs(losses[k].item() - v) < loss_tol * max(1.0, abs(v)), (k, losses[k].item(), v)
    sum(losses.values()).backward()
    params = dict(m.named_parameters())
    for k in [f for f in g.files if f.startswith("train/grad/")]:
        name = k.split("/", 2)[2]
        ref = torch.tensor(g[k])
        got = params[name].grad.float().cpu()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        assert cos > cos_min, (name, cos)
        ratio = (got.norm() / ref.norm()).item()
        assert 0.9 < ratio < 1.1, (name, ratio)


def test_train_step_matches_reference_s320_fp32(cuda):
    """D-FINE-s (the model of BASELINE configs[1], fp32): losses 2e-3, gradient direction 0.9999 vs the reference."""
    _train_step_vs_golden(cuda, "s", "model_s320.npz", amp=False, loss_tol=2e-3, cos_min=0.9999)


def test_config2_s640_fp32_train_step_properties(cuda, monkeypatch):
    """BASELINE configs[1]: D-FINE-s 640x640 bs=16 fp32 on one MI355X.  Full size -> properties: finite losses, every
    parameter moves, the fused HIP criterion equals the torch composition on the same outputs, valid assignments."""
    import bench
    from custom_d_fine_amd.dl.synthetic import make_batch
    step = bench.build_step("s", 640, cuda, None)
    images, targets = make_batch(16, 640, seed=42, device=cuda)
    before = step.fused.flat_param.detach().clone()
    loss, loss_dict = step(images, targets)
    assert torch.isfinite(loss) and len(loss_dict) == 38
    assert all(torch.isfinite(v) for v in loss_dict.values())
    after = step.fused.flat_param
    assert torch.isfinite(after).all() and (after != before).float().mean().item() > 0.9
    model, crit = step.model, step.criterion
    with torch.no_grad():
        out = model(images, targets)
    fused = crit(out, targets)
    monkeypatch.setattr(type(crit), "_fusable", lambda self, outputs: False)
    plain = crit(out, targets)
    assert set(fused) == set(plain)
    for k in plain:
        a, b = fused[k].item(), plain[k].item()
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (k, a, b)
    for (rows, cols), t in zip(crit.matcher(out, targets)["indices"], targets):
        n = len(t["labels"])
        assert len(rows) == len(cols) == min(n, 300) and sorted(cols.tolist()) == list(range(n))


@pytest.mark.parametrize("seed", [0, 1])
def test_criterion_on_gpu_matches_reference(cuda, seed):
    g = np.load(f"{G}/criterion.npz")
    crit = dfine.build_loss("s", 6, 0.0, False)
    outputs = helpers.make_criterion_outputs(seed, device=cuda)
    targets, meta = helpers.criterion_targets_and_meta(device=cuda)
    outputs["dn_meta"] = meta
    losses = crit(outputs, targets)
    want = {k.split("/", 2)[2]: float(g[k]) for k in g.files if k.startswith(f"s{seed}/loss/")}
    assert set(losses) == set(want)
    for k, v in want.items():
        assert abs(losses[k].item() - v) < 1e-4 * max(1.0, abs(v)), (k, losses[k].item(), v)
    sum(losses.values()).backward()
    np.testing.assert_allclose(outputs["pred_logits"].grad.cpu().numpy(), g[f"s{seed}/grad/pred_logits"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(outputs["pred_corners"].grad.cpu().numpy(), g[f"s{seed}/grad/pred_corners"], rtol=1e-3, atol=1e-5)


def test_bf16_autocast_train_step_runs_and_is_close(cuda):
    """bf16 is the throughput mode (BASELINE configs[2]); checked against the fp32 run of the same
    model with a looser, separately stated tolerance: total loss within 3 %."""
    m = dfine.build_model("n", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    crit = dfine.build_loss("n", 80, 0.0, False)
    targets = helpers.make_targets(2, 80, device=cuda)
    x = helpers.make_images(2, 320).to(cuda)
    totals = []
    for amp in (False, True):
        U.set_denoising_generator(torch.Generator().manual_seed(11))
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = m(x, targets)
        finally:
            U.set_denoising_generator(None)
        loss = sum(crit(out, targets).values())
        loss.backward()
        totals.append(loss.item())
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        m.zero_grad()
    assert abs(totals[0] - totals[1]) < 0.03 * totals[0], totals


def test_full_size_train_step_properties(cuda, monkeypatch):
    """BASELINE.json's bench configuration (D-FINE-m, 640x640, batch 32, bf16): the CPU oracle takes minutes there,
    so the full-size check is through properties - finite losses / gradients, every parameter updated, the fused HIP
    criterion equal to the torch composition (the reference's formulas, on the GPU) on the SAME model outputs, and
    matcher indices that are valid one-to-one assignments."""
    import bench
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.dl.synthetic import make_batch
    kernels.reload_env()
    try:
        step = bench.build_step("m", 640, cuda, torch.bfloat16)
        images, targets = make_batch(32, 640, seed=42, device=cuda)
        before = step.fused.flat_param.detach().clone()
        loss, loss_dict = step(images, targets)
        assert torch.isfinite(loss) and len(loss_dict) >= 38
        assert all(torch.isfinite(v) for v in loss_dict.values())
        after = step.fused.flat_param
        assert torch.isfinite(after).all() and (after != before).float().mean().item() > 0.9   # zero-init heads keep zeros

        model, crit = step.model, step.criterion
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(images, targets)
        fused = crit(out, targets)
        monkeypatch.setattr(type(crit), "_fusable", lambda self, outputs: False)
        plain = crit(out, targets)                           # torch composition of the same losses, fp32, on the GPU
        assert set(fused) == set(plain)
        for k in plain:
            a, b = fused[k].item(), plain[k].item()
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (k, a, b)

        heads = [{k: v for k, v in out.items() if "aux" not in k}] + list(out["aux_outputs"])
        for m in crit.matcher.match_heads(heads, targets):
            for (rows, cols), t in zip(m, targets):
                n = len(t["labels"])
                assert len(rows) == len(cols) == min(n, 300)
                assert len(set(rows.tolist())) == len(rows) and sorted(cols.tolist()) == list(range(n))
    finally:
        kernels.reload_env()


def test_backbone_encoder_m320_vs_reference(cuda):
    """HGNetv2 + HybridEncoder of D-FINE-m in fp32 against the reference's features and parameter gradients
    (3e-3 of the feature scale; gradient cosine 0.9998, norm 3e-3; fixtures are fp16 slices).  The early backbone stages
    amplify rounding-order differences (batch statistics over two images): measured 1 - cos = 3-5e-5 with MIOpen's fp32
    convolutions and 9-11e-5 with the f32-MFMA kernels of csrc/conv_f32.hip (themselves within 3e-7 of fp64 convolutions on
    every layer shape, tests/test_conv_f32_gpu.py), 4e-8 from stage 3 on with either; the features agree to 3.8e-4."""
    g = np.load(f"{G}/backbone_encoder_m320.npz")
    m = dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    x = helpers.make_images(2, 320).to(cuda)
    feats = m.encoder(m.backbone(x))
    loss = 0
    for i, f in enumerate(feats):
        ref = torch.tensor(g[f"feat{i}"].astype(np.float32))
        got = f.detach().float().cpu()[:1]
        assert (got - ref).abs().max() <= 3e-3 * ref.abs().max(), (i, (got - ref).abs().max().item())
        loss = loss + (f.float() * helpers.make_cotangent(f.shape, 50 + i).to(cuda)).sum()
    loss.backward()
    params = dict(m.named_parameters())
    for k in helpers.BACKBONE_ENCODER_GRAD_KEYS:
        ref = torch.tensor(g[f"grad/{k}"].astype(np.float32)) * float(g[f"gscale/{k}"])
        got = helpers.compact_rows(params[k].grad.float().cpu())
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        ratio = (got.norm() / ref.norm()).item()
        assert cos > 0.9998, (k, cos)
        assert abs(ratio - 1) < 3e-3, (k, ratio)


def test_bf16_blocks_vs_fp32_blocks_m320(cuda):
    """The bf16 (MFMA) path block by block.  End to end this network is not a usable bf16 anchor - with the seeded random
    weights and batch statistics over 2 images, plain ATen bf16 autocast already decorrelates the encoder features to cosine
    0.7-0.85 of the fp32 ones (and the full model adds top-k selection and the matcher) - so every backbone / encoder block
    is run in bf16 autocast on the inputs it saw in the fp32 run above (itself pinned to the reference), and its output,
    input gradient and parameter gradients are compared with the block's own fp32 results: cosine >= 0.999 / 0.97 / 0.9 (a
    block is up to 8 conv + batch-statistics BN units deep; the kernels themselves are pinned tightly by their unit tests)."""
    import torch.nn as nn
    m = dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    blocks = [("backbone.stem", m.backbone.stem)]
    for si, st in enumerate(m.backbone.stages):
        if hasattr(st, "downsample") and not isinstance(st.downsample, nn.Identity):
            blocks.append((f"backbone.stages.{si}.downsample", st.downsample))
        blocks += [(f"backbone.stages.{si}.blocks.{bi}", b) for bi, b in enumerate(st.blocks)]
    enc = m.encoder
    for name in ("lateral_convs", "fpn_blocks", "downsample_convs", "pan_blocks"):     # (input_proj units are called through kernels.conv_bn_act, no module hook)
        blocks += [(f"encoder.{name}.{i}", b) for i, b in enumerate(getattr(enc, name))]
    captured = {}
    def grab(n):
        def hook(mod, inp, out):
            x0 = inp[0]         # FPN / PAN blocks take the fusion inputs as a list (concatenation read in place)
            captured[n] = ([t.detach() for t in x0] if isinstance(x0, (list, tuple)) else x0.detach(), out.detach())
        return hook
    hooks = [b.register_forward_hook(grab(n)) for n, b in blocks]
    x = helpers.make_images(2, 320).to(cuda)
    with torch.no_grad():
        m.encoder(m.backbone(x))
    for h in hooks:
        h.remove()
    assert len(captured) == len(blocks)
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0).item()
    worst = {}
    for n, b in blocks:
        xin, _ = captured[n]
        res = []
        for amp in (False, True):
            image_in = n == "backbone.stem"          # the network input: no gradient (a train step never asks for one)
            if isinstance(xin, list):
                xi = [t.clone().requires_grad_(True) for t in xin]
            else:
                xi = xin.clone().requires_grad_(not image_in)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                y = b(xi)
            go = helpers.make_cotangent(y.shape, 77).to(cuda)
            b.zero_grad()
            (y.float() * go).sum().backward()
            gx = torch.cat([t.grad.float().flatten() for t in xi]) if isinstance(xi, list) else (
                torch.ones(1, device=cuda) if image_in else xi.grad.detach())
            res.append((y.detach(), gx, {k: p.grad.detach().clone() for k, p in b.named_parameters() if p.grad is not None}))
        (y0, gx0, gp0), (y1, gx1, gp1) = res
        cy, cx = cos(y0, y1), cos(gx0, gx1)
        cp = min((cos(gp0[k], gp1[k]) for k in gp0 if gp0[k].numel() > 16 and gp0[k].abs().max() > 0), default=1.0)
        worst[n] = (cy, cx, cp)
    bad = {n: v for n, v in worst.items() if v[0] < 0.999 or v[1] < 0.97 or v[2] < 0.9}
    assert not bad, bad


def bf16_block_parity_table(cuda):
    """Every backbone / encoder block of D-FINE-m (seeded weights, 2 images, 320x320) run three ways on the inputs it saw in the
    fp32 forward: fp32, bf16 autocast on the HIP kernels, bf16 autocast composed from ATen ops (MIOpen convolutions,
    at::batch_norm: `DFINE_HIP_UNITS=0` and friends).  -> {block: {"hip": (1-cos y, 1-cos dx, worst 1-cos dparam, its name),
    "aten": (...)}} with the cosines taken against the fp32 results."""
    import torch.nn as nn
    from custom_d_fine_amd import kernels
    m = dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    blocks = [("backbone.stem", m.backbone.stem)]
    for si, st in enumerate(m.backbone.stages):
        if hasattr(st, "downsample") and not isinstance(st.downsample, nn.Identity):
            blocks.append((f"backbone.stages.{si}.downsample", st.downsample))
        blocks += [(f"backbone.stages.{si}.blocks.{bi}", b) for bi, b in enumerate(st.blocks)]
    enc = m.encoder
    for name in ("lateral_convs", "fpn_blocks", "downsample_convs", "pan_blocks"):
        blocks += [(f"encoder.{name}.{i}", b) for i, b in enumerate(getattr(enc, name))]
    captured = {}

    def grab(n):
        def hook(mod, inp, out):
            x0 = inp[0]
            captured[n] = [t.detach() for t in x0] if isinstance(x0, (list, tuple)) else x0.detach()
        return hook
    hooks = [b.register_forward_hook(grab(n)) for n, b in blocks]
    with torch.no_grad():
        m.encoder(m.backbone(helpers.make_images(2, 320).to(cuda)))
    for h in hooks:
        h.remove()
    switches = ("DFINE_HIP_UNITS", "DFINE_MFMA_CONV", "DFINE_STEM", "DFINE_SEG_CONV", "DFINE_BN2", "DFINE_DUAL_CONV",
                "DFINE_HIP_LINEAR", "DFINE_LN_FUSED", "DFINE_HIP_ATTN")
    saved = {s: os.environ.get(s) for s in switches}

    def run(b, xin, amp):
        image_in = b is m.backbone.stem              # the network input: no gradient (a train step never asks for one)
        xi = [t.clone().requires_grad_(True) for t in xin] if isinstance(xin, list) else xin.clone().requires_grad_(not image_in)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = b(xi)
        b.zero_grad()
        (y.float() * helpers.make_cotangent(y.shape, 77).to(cuda)).sum().backward()
        gx = torch.cat([t.grad.float().flatten() for t in xi]) if isinstance(xi, list) else (
            torch.ones(1, device=cuda) if image_in else xi.grad.detach())
        return y.detach(), gx, {k: p.grad.detach().clone() for k, p in b.named_parameters() if p.grad is not None}

    results = {}
    try:
        # ATen's bf16 composition is not reproducible from run to run (atomically accumulated MIOpen / at::batch_norm
        # gradients: the worst BatchNorm weight gradient of one block moved between 6.2e-2 and 7.8e-2 over a dozen runs while the
        # HIP path returned 8.32e-2 every time): its distance is taken as the largest of three runs
        for mode in ("fp32", "hip", "aten", "aten2", "aten3"):
            for s in switches:
                if mode.startswith("aten"):
                    os.environ[s] = "0"
                else:
                    os.environ.pop(s, None)
            kernels.reload_env()
            for n, b in blocks:
                results[(mode, n)] = run(b, captured[n], mode != "fp32")
    finally:
        for s, v in saved.items():
            if v is None:
                os.environ.pop(s, None)
            else:
                os.environ[s] = v
        kernels.reload_env()
    one_minus_cos = lambda a, b: 1.0 - torch.nn.functional.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0).item()
    table = {}
    for n, _ in blocks:
        y0, gx0, gp0 = results[("fp32", n)]
        row = {}
        for mode in ("hip", "aten", "aten2", "aten3"):
            y1, gx1, gp1 = results[(mode, n)]
            assert gp0.keys() == gp1.keys(), (n, mode)
            wk, wp = max(((k, one_minus_cos(gp0[k], gp1[k])) for k in gp0 if gp0[k].numel() > 16 and gp0[k].abs().max() > 0),
                         key=lambda t: t[1], default=("", 0.0))
            row[mode] = (one_minus_cos(y0, y1), one_minus_cos(gx0, gx1), wp, wk)
        reps = [row.pop("aten2"), row.pop("aten3"), row["aten"]]
        row["aten"] = (max(r[0] for r in reps), max(r[1] for r in reps), max(r[2] for r in reps), row["aten"][3])
        table[n] = row
    return table


def test_bf16_hip_blocks_no_worse_than_aten_bf16_blocks_m320(cuda):
    """The tight anchor of the bf16 path.  Two bf16 implementations of one block cannot be compared with each other at
    cosine 0.999 - measured: the BatchNorm weight gradients (sums of dy * xhat with cancelling signs over batch statistics of
    two images) of the HIP path and of the ATen bf16 composition differ by cosine 0.90-0.98 although both round to bf16 at the
    same points.  What can be demanded is that the HIP path is AS CLOSE TO fp32 AS ATen's bf16 path: per block, the distance
    1 - cos to the block's fp32 output / input gradient / worst parameter gradient may exceed ATen-bf16's by at most a factor
    1.6 (+ 2e-4 absolute; the stem block - all of it on the HIP kernels at 320 x 320 since round 4, its stride-2 data gradient
    included - measures 1.55 on the BatchNorm bias gradient of stem2b, everything else below 1.3).  A wrong tap, a dropped weight-gradient split or a mis-scaled statistic moves the HIP distance by
    orders of magnitude, ATen's not at all (the fp32 comparison above only bounds it by 0.1)."""
    table = bf16_block_parity_table(cuda)
    bad = {}
    for n, row in table.items():
        for qi, q in enumerate(("y", "dx", "dparam")):
            h, a = row["hip"][qi], row["aten"][qi]
            if h > 1.6 * a + 2e-4:
                bad[(n, q)] = (round(h, 5), round(a, 5), row["hip"][3] if q == "dparam" else "")
    assert not bad, bad


def bf16_decoder_parity_table(cuda):
    """The decoder side of D-FINE-m (seeded weights, 2 images, 320 x 320, denoising group from the targets) run three ways on
    the inputs captured in an fp32 train forward: fp32, bf16 autocast on the HIP kernels, bf16 autocast on ATen (F.linear,
    SDPA, nn.LayerNorm; the deformable gather has no ATen form on the GPU and stays).  Blocks: every TransformerDecoderLayer,
    the query selection (`_get_decoder_input`: enc_output + score / box heads on the selected rows) and the whole decoder
    stack with its FDR heads (boxes / logits / corners of all layers).
    -> {block: {"hip": (1-cos y, 1-cos dx, worst 1-cos dparam, name), "aten": (...)}} against the fp32 results."""
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.d_fine.arch import utils as U
    m = dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda).train()
    dec = m.decoder
    targets = helpers.make_targets(2, 80, device=cuda)
    captured = {}

    def pre(name):
        def hook(mod, args, kwargs):
            captured[name] = (tuple(a.detach() if torch.is_tensor(a) else a for a in args),
                              {k: (v.detach() if torch.is_tensor(v) else v) for k, v in kwargs.items()})
        return hook
    hooks = [layer.register_forward_pre_hook(pre(f"decoder.layers.{i}"), with_kwargs=True) for i, layer in enumerate(dec.decoder.layers)]
    hooks.append(dec.decoder.register_forward_pre_hook(pre("decoder.stack"), with_kwargs=True))
    orig_gdi = dec._get_decoder_input

    orig_eo = dec._enc_output
    memory_seen = {}

    def spy_gdi(memory, spatial_shapes, dn_logits=None, dn_boxes=None):
        memory_seen["memory"], memory_seen["shapes"] = memory.detach(), spatial_shapes
        return orig_gdi(memory, spatial_shapes, dn_logits, dn_boxes)

    def spy_eo(t):
        if torch.is_grad_enabled():          # the differentiable call: the 300 selected (masked) rows
            captured["query_heads"] = ((t.detach(),), {})
        return orig_eo(t)
    dec._get_decoder_input, dec._enc_output = spy_gdi, spy_eo
    U.set_denoising_generator(torch.Generator().manual_seed(11))
    try:
        m(helpers.make_images(2, 320).to(cuda), targets)
    finally:
        U.set_denoising_generator(None)
        dec._get_decoder_input, dec._enc_output = orig_gdi, orig_eo
        for h in hooks:
            h.remove()

    def float_leaves(args):
        return [a for a in args if torch.is_tensor(a) and a.is_floating_point()]

    def run(name, amp):
        args, kwargs = captured[name]
        args = tuple(a.clone().requires_grad_(True) if (torch.is_tensor(a) and a.is_floating_point()) else a for a in args)
        dec.zero_grad()
        # (cache_enabled=False: ATen's F.linear under autocast would reuse a bf16 weight copy that a no-grad call made)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp, cache_enabled=False):
            if name == "query_heads":
                top_mem = orig_eo(args[0])
                outs = [top_mem, dec._enc_scores(top_mem), dec.enc_bbox_head(top_mem)]
            elif name == "decoder.stack":
                res = dec.decoder(*args, **kwargs)
                # boxes, logits, corners of every layer (per-layer lists since round 6) + the pre heads
                outs = [torch.stack(r) if isinstance(r, (list, tuple)) else r for r in (res[0], res[1], res[2])] + [res[4], res[5]]
            else:
                outs = [dec.decoder.layers[int(name.rsplit(".", 1)[1])](*args, **kwargs)]
        loss = sum((o.float() * helpers.make_cotangent(o.shape, 31 + i).to(cuda)).sum() for i, o in enumerate(outs))
        loss.backward()
        leaves = [a for a in float_leaves(args) if a.grad is not None]
        gx = torch.cat([a.grad.float().flatten() for a in leaves])
        gp = {k: p.grad.detach().clone() for k, p in dec.named_parameters() if p.grad is not None}
        return torch.cat([o.detach().float().flatten() for o in outs]), gx, gp

    switches = ("DFINE_HIP_LINEAR", "DFINE_LN_FUSED", "DFINE_HIP_ATTN")
    saved = {s: os.environ.get(s) for s in switches}
    results, topk = {}, {}
    try:
        for mode in ("fp32", "hip", "aten", "aten2", "aten3"):
            for s in switches:
                if mode.startswith("aten"):
                    os.environ[s] = "0"
                else:
                    os.environ.pop(s, None)
            kernels.reload_env()
            for name in captured:
                results[(mode, name)] = run(name, mode != "fp32")
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32", cache_enabled=False):
                mem = memory_seen["memory"]
                _, valid = dec._anchors_for(memory_seen["shapes"], mem.device)
                topk[mode] = dec._topk_indices(dec._enc_scores(dec._enc_output(valid.to(mem.dtype) * mem)), dec.num_queries)
    finally:
        for s, v in saved.items():
            if v is None:
                os.environ.pop(s, None)
            else:
                os.environ[s] = v
        kernels.reload_env()
    omc = lambda a, b: 1.0 - torch.nn.functional.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0).item()
    table = {}
    for name in captured:
        y0, gx0, gp0 = results[("fp32", name)]
        row = {}
        for mode in ("hip", "aten", "aten2", "aten3"):
            y1, gx1, gp1 = results[(mode, name)]
            assert gp0.keys() == gp1.keys(), (name, mode, set(gp0) ^ set(gp1))
            wk, wp = max(((k, omc(gp0[k], gp1[k])) for k in gp0 if gp0[k].numel() > 16 and gp0[k].abs().max() > 0),
                         key=lambda t: t[1], default=("", 0.0))
            row[mode] = (omc(y0, y1), omc(gx0, gx1), wp, wk)
        reps = [row.pop("aten2"), row.pop("aten3"), row["aten"]]
        row["aten"] = (max(r[0] for r in reps), max(r[1] for r in reps), max(r[2] for r in reps), row["aten"][3])
        table[name] = row

    def overlap(a, b):
        return min(len(set(x.tolist()) & set(y.tolist())) / len(x) for x, y in zip(a, b))
    table["query_topk"] = {"hip": overlap(topk["hip"], topk["fp32"]),
                           "aten": max(overlap(topk[m_], topk["fp32"]) for m_ in ("aten", "aten2", "aten3"))}
    return table


def test_bf16_hip_decoder_blocks_no_worse_than_aten_bf16_m320(cuda):
    """The bf16 anchor of the decoder side of the headline configuration: every decoder layer, the query selection and the
    decoder stack with its FDR heads are as close to their fp32 results as the ATen bf16 composition of the same block
    (1 - cos to fp32 at most 1.5 x ATen's + 2e-4; same criterion as the backbone / encoder blocks above), and close in absolute
    terms (output 1 - cos < 2e-3: a token-stream block has no batch statistics to amplify bf16 rounding)."""
    table = bf16_decoder_parity_table(cuda)
    assert {"query_heads", "query_topk", "decoder.stack", "decoder.layers.0", "decoder.layers.3"} <= set(table)
    sel = table.pop("query_topk")
    # the bf16 selection shares as many of its 300 queries with the fp32 selection as ATen's bf16 selection does (- 2 %)
    assert sel["hip"] >= sel["aten"] - 0.02 and sel["hip"] > 0.5, sel
    bad = {}
    for n, row in table.items():
        for qi, q in enumerate(("y", "dx", "dparam")):
            h, a = row["hip"][qi], row["aten"][qi]
            if h > 1.5 * a + 2e-4:
                bad[(n, q)] = (round(h, 6), round(a, 6), row["hip"][3] if q == "dparam" else "")
        if row["hip"][0] > 2e-3:
            bad[(n, "y abs")] = row["hip"][0]
    assert not bad, bad
