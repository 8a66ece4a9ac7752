"""The bf16 anchor of the HEADLINE configuration (BASELINE.json configs[2]: D-FINE-m, 640 x 640, batch 32).

The fp32 HIP path is pinned to the reference by goldens (model_n320 / model_s320 / backbone_encoder_m320 / model_m640_eval_b3,
tests/test_model_gpu.py); the bf16 path of the bench was only checked block by block at 320 x 320 and, at full size, against
itself.  Here the SAME weights, batch and denoising noise run one forward + criterion + backward at full size three times:

  A  fp32 math (f32-MFMA kernels, f32 atomics in the deformable-attention backward)          - the golden-pinned path
  B  bf16 autocast, d(value) of the deformable attention accumulated in packed f16 pairs      - what bench.py times
  C  bf16 autocast, d(value) accumulated with f32 atomics (hip.MSDA_ACC_MODE = 0)             - the reference's accumulate type

The step has two DISCRETE decisions - the top-300 anchor selection (`_topk_indices`) and the Hungarian assignments - and at
initialisation both are ill-conditioned: fp32 math on images merely ROUNDED to bf16 (a 2^-9 relative change of the input) already
moves the backbone / encoder gradients to cosine 0.24 / 0.31 of the unperturbed fp32 ones (ATen's bf16 composition: 0.08 / 0.14,
the HIP bf16 path: 0.06 / 0.12; measured by this file with DFINE_ANCHOR_PRINT=1).  The gradient comparison therefore runs with
both decisions FROZEN to the fp32 run's (recorded in A, replayed in B and C): what is left is a smooth function of the
arithmetic.  The loss VALUES are also compared with the decisions free (the mode bench.py times).

and the test asserts B against A (every loss term, the gradient of every top-level module as cosine / norm ratio) and B against
C (what the reduced-precision accumulate costs: nothing that shows in the gradients).  Bounds are ~2 x the values measured on
an MI355X, recorded next to each assertion; a kernel change that moves bf16 away from fp32, or the f16 accumulate away from the
f32 one, fails here.
"""
import os

import pytest
import torch

from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.d_fine.arch import utils as U

pytestmark = pytest.mark.gpu


class _Decisions:
    """Records (mode 'record') or replays (mode 'replay') the step's discrete decisions: top-k anchor indices and assignments."""

    def __init__(self, model, crit):
        self.dec, self.matcher = model.decoder, crit.matcher
        self.topk, self.match, self.mode = None, None, None
        topk0, match0 = self.dec._topk_indices, self.matcher.match_heads_device

        def topk(logits, k):
            if self.mode == "replay":
                return self.topk.clone()
            ind = topk0(logits, k)
            if self.mode == "record":
                self.topk = ind.clone()
            return ind

        def match(heads, targets):
            if self.mode == "replay":
                cols, off, sizes = self.match
                return cols.clone(), off, sizes
            out = match0(heads, targets)
            assert out is not None, "the device-plan path is the one bench.py runs"
            if self.mode == "record":
                self.match = (out[0].clone(), out[1], out[2])
            return out

        self.dec._topk_indices, self.matcher.match_heads_device = topk, match


def _run(model, crit, images, targets, amp, acc_mode, decisions=None, mode=None):
    from custom_d_fine_amd import hip
    old = hip.MSDA_ACC_MODE
    hip.MSDA_ACC_MODE = acc_mode
    U.set_denoising_generator(torch.Generator().manual_seed(11))
    if decisions is not None:
        decisions.mode = mode
    try:
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = model(images, targets)
        with torch.autocast("cuda", enabled=False):
            ld = crit(out, targets)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
    finally:
        U.set_denoising_generator(None)
        hip.MSDA_ACC_MODE = old
    losses = {k: v.item() for k, v in ld.items()}
    grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    return losses, grads


def _group_stats(ga, gb):
    """per top-level module: (cosine, |a| / |b|) of the concatenated parameter gradients"""
    out = {}
    for grp in ("backbone", "encoder", "decoder"):
        keys = [k for k in gb if k.startswith(grp + ".") and k in ga]
        a = torch.cat([ga[k].flatten() for k in keys]).double()
        b = torch.cat([gb[k].flatten() for k in keys]).double()
        out[grp] = (torch.nn.functional.cosine_similarity(a, b, dim=0).item(), (a.norm() / b.norm()).item())
    return out


def test_bf16_train_step_m640_bs32_against_fp32_and_f32_accumulate(cuda):
    from custom_d_fine_amd.dl.synthetic import make_batch
    torch.manual_seed(0)
    model = dfine.build_model("m", 80, False, str(cuda), img_size=[640, 640]).train()
    crit = dfine.build_loss("m", 80, 0.0, False)
    images, targets = make_batch(32, 640, seed=42, device=cuda)
    dec = _Decisions(model, crit)
    la, ga = _run(model, crit, images, targets, False, -1, dec, "record")        # A: fp32, decisions recorded
    lf, _ = _run(model, crit, images, targets, True, -1, dec, None)              # bf16 with its OWN decisions (bench mode): losses only
    lb, gb = _run(model, crit, images, targets, True, -1, dec, "replay")         # B: bf16, f16-pair accumulate, A's decisions
    lc, gc = _run(model, crit, images, targets, True, 0, dec, "replay")          # C: bf16, f32 atomics, A's decisions
    lb2, gb2 = _run(model, crit, images, targets, True, -1, dec, "replay")       # B again: run-to-run noise of the atomics
    assert set(la) == set(lb) == set(lc) == set(lf) and len(la) == 48
    tot = {k: sum(v.values()) for k, v in (("a", la), ("b", lb), ("c", lc), ("b2", lb2), ("free", lf))}
    rel = lambda x, y: {k: abs(x[k] - y[k]) / max(abs(y[k]), 0.05) for k in y}      # noqa: E731
    rel_ba, rel_bc, rel_fa = rel(lb, la), rel(lb, lc), rel(lf, la)
    s_ba, s_bc, s_bb = _group_stats(gb, ga), _group_stats(gb, gc), _group_stats(gb, gb2)
    # E: fp32 math on images ROUNDED to bf16 (a 2^-9 relative change of the input, nothing else) - how stable is the quantity
    # being compared?   D: bf16 autocast composed from ATen / library ops (MIOpen, hipBLASLt, SDPA) instead of the HIP kernels
    le, ge = _run(model, crit, images.bfloat16().float(), targets, False, -1, dec, "replay")
    s_ea = _group_stats(ge, ga)
    from custom_d_fine_amd import kernels
    switches = ("DFINE_HIP_UNITS", "DFINE_MFMA_CONV", "DFINE_STEM", "DFINE_SEG_CONV", "DFINE_BN2", "DFINE_DUAL_CONV",
                "DFINE_HIP_LINEAR", "DFINE_LN_FUSED", "DFINE_HIP_ATTN")
    try:
        for s_ in switches:
            os.environ[s_] = "0"
        os.environ["DFINE_ALLOW_LIBRARY"] = "1"
        kernels.reload_env()
        ld, gd = _run(model, crit, images, targets, True, -1, dec, "replay")
    finally:
        for s_ in switches + ("DFINE_ALLOW_LIBRARY",):
            os.environ.pop(s_, None)
        kernels.reload_env()
    s_da = _group_stats(gd, ga)
    if os.environ.get("DFINE_ANCHOR_PRINT") == "1":
        top = lambda d: sorted(d.items(), key=lambda kv: -kv[1])[:4]              # noqa: E731
        print("\ntotals", tot)
        print("worst loss terms bf16 vs fp32, frozen decisions:", top(rel_ba))
        print("worst loss terms bf16 vs fp32, free decisions:", top(rel_fa))
        print("worst loss terms f16-acc vs f32-acc:", top(rel_bc))
        print("gradients bf16 vs fp32, frozen decisions (cos, norm ratio):", s_ba)
        print("gradients f16-acc vs f32-acc:", s_bc)
        print("gradients bf16 vs bf16 again:", s_bb)
        print("fp32 on bf16-rounded images vs fp32, frozen:", s_ea, sum(le.values()))
        print("ATen bf16 vs fp32, frozen:", s_da, sum(ld.values()))
        print("HIP bf16 vs ATen bf16, frozen:", _group_stats(gb, gd))
    assert all(torch.isfinite(torch.tensor(list(v.values()))).all() for v in (la, lb, lc, lf))
    # ---- the bf16 step against the golden-pinned fp32 step: loss values with free and with frozen decisions
    assert abs(tot["free"] - tot["a"]) <= BOUND["total_rel"] * abs(tot["a"]), tot
    assert abs(tot["b"] - tot["a"]) <= BOUND["total_rel"] * abs(tot["a"]), tot
    assert max(rel_fa.values()) <= BOUND["term_rel_free"], max(rel_fa.items(), key=lambda kv: kv[1])
    assert max(rel_ba.values()) <= BOUND["term_rel"], max(rel_ba.items(), key=lambda kv: kv[1])
    # ---- gradients, frozen decisions.  The decoder's gradient is a stable quantity (0.986 under the input rounding E) and the
    # bf16 step reproduces it (0.973); the DIRECTION of the backbone / encoder gradients at initialisation is not - fp32 itself
    # keeps only 0.42 / 0.60 of it under E (train-mode BatchNorm over ~130 conv units + ReLU kinks) - so for those two groups the
    # assertion is relative: the HIP bf16 path stays as close to fp32 as the library (ATen) bf16 composition does
    # (measured 0.151 / 0.313 against 0.133 / 0.281), above a floor of half the measured value, and every group's gradient
    # NORM agrees with fp32 within 3 % (measured 0.1-0.7 %)
    for grp, (cos, ratio) in s_ba.items():
        assert cos >= BOUND["cos"][grp], (grp, cos, ratio)
        assert cos >= s_da[grp][0] - BOUND["aten_margin"], (grp, cos, s_da[grp])
        assert abs(ratio - 1.0) <= BOUND["norm"], (grp, cos, ratio)
    assert s_ea["decoder"][0] > 0.97 and s_ea["backbone"][0] < 0.9, s_ea      # the calibration itself (fp32 under input rounding)
    # ---- B against C: packed-f16 accumulate of d(value) against f32 atomics, everything else equal (same forward kernels on
    # the same inputs: the loss terms differ by nothing but the matcher-independent summation order)
    assert max(rel_bc.values()) <= BOUND["acc_term_rel"], max(rel_bc.items(), key=lambda kv: kv[1])
    for grp, (cos, ratio) in s_bc.items():
        # no further from the f32-accumulate gradients than two runs of the SAME configuration are from each other, plus a margin
        noise = 1.0 - s_bb[grp][0]
        assert 1.0 - cos <= BOUND["acc_cos_margin"] + 3.0 * noise, (grp, cos, ratio, noise)
        assert abs(ratio - 1.0) <= BOUND["acc_norm"], (grp, cos, ratio)


# Measured on an MI355X (round 5, DFINE_ANCHOR_PRINT=1):
#   totals: fp32 41.4151, bf16 frozen 41.4295 (+0.035 %), bf16 free 41.5411 (+0.30 %); f16-acc vs f32-acc totals equal to 1e-8
#   worst loss term bf16 vs fp32: frozen 1.95 % (loss_vfl_aux_1), free 6.3 % (loss_vfl_enc_0: other anchors selected)
#   gradients bf16 vs fp32, frozen (cos, |g| ratio): backbone 0.151 / 1.007, encoder 0.313 / 1.006, decoder 0.973 / 0.9995
#   ATen bf16 vs fp32, frozen: backbone 0.133 / 0.996, encoder 0.281 / 1.003, decoder 0.970 / 0.998
#   fp32 on bf16-rounded images vs fp32, frozen: backbone 0.421, encoder 0.602, decoder 0.986
#   f16-acc vs f32-acc: loss terms 6e-7, cos backbone 0.99964 / encoder 0.99989 / decoder 1.0 = the run-to-run noise of B itself
#   (0.99965 / 0.99989 / 1.0), norm ratios within 3.4e-4
BOUND = {
    "total_rel": 0.01, "term_rel": 0.05, "term_rel_free": 0.15,
    "cos": {"backbone": 0.08, "encoder": 0.16, "decoder": 0.95}, "aten_margin": 0.05, "norm": 0.03,
    "acc_term_rel": 1e-4, "acc_cos_margin": 1e-3, "acc_norm": 5e-3,
}


def test_bf16_blocks_m640_bs32_vjp_against_fp32(cuda):
    """A gradient anchor for the headline configuration that CAN fail (review r5 item 7).

    End to end, bf16 and fp32 cannot be compared in backbone / encoder at initialisation: the relative difference of the FEATURES
    grows by ~1.2 x per conv + BatchNorm unit (ReLU outputs are dominated by their mean, which the next BatchNorm removes - the
    rounding noise it does not) and reaches 0.45 - 0.70 at the encoder outputs, in training mode and with frozen statistics alike
    (tools/probe/eval_bf16_drift.py; profiles/r06_bf16_anchor.txt) - so neither the train-step comparison above nor a
    frozen-statistics pull-back through the whole stack can tell a wrong gradient from rounding.  What CAN be compared is every
    block on its own: D-FINE-m, 640 x 640, batch 32, one fp32 forward records the inputs of every StemBlock / HG_Block /
    RepNCSPELAN4 / SCDown / AIFI layer; each block then runs forward + backward (training mode, batch statistics, a FIXED
    cotangent) in fp32 on the HIP f32 path (golden-pinned: backbone_encoder_m320) and in bf16 on the benched kernels, from the SAME
    inputs.  Three to fourteen units deep, the amplification stays small, and a wrong data gradient, weight gradient or
    BatchNorm backward of any kernel at the benched shapes shows as a cosine well below the bounds asserted here
    (measured values: profiles/r06_bf16_anchor.txt; printed with DFINE_ANCHOR_PRINT=1)."""
    from custom_d_fine_amd.dl.synthetic import make_batch
    torch.manual_seed(0)
    model = dfine.build_model("m", 80, False, str(cuda), img_size=[640, 640]).train()
    images, _ = make_batch(32, 640, seed=42, device=cuda)
    body = torch.nn.Sequential(model.backbone, model.encoder)
    kinds = ("StemBlock", "HG_Block", "RepNCSPELAN4", "SCDown", "TransformerEncoderLayer")
    blocks = [(n, m) for n, m in body.named_modules() if type(m).__name__ in kinds]
    assert len(blocks) >= 12
    rec = {}

    def keep(t):
        if torch.is_tensor(t):
            return t.detach().float().clone()
        if isinstance(t, (list, tuple)):
            return [keep(u) for u in t]
        return t

    hooks = [m.register_forward_pre_hook((lambda name: lambda mod, args, kwargs: rec.__setitem__(name, (keep(args), keep(dict(kwargs)))))(n),
                                         with_kwargs=True) for n, m in blocks]
    with torch.no_grad():
        body(images)                                     # fp32: the inputs every block sees
    for h in hooks:
        h.remove()

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()

    def leaves(t, dtype, grad=True):
        if torch.is_tensor(t):
            return t.detach().to(dtype, copy=True).requires_grad_(grad) if t.is_floating_point() else t
        if isinstance(t, list):
            return [leaves(u, dtype, grad) for u in t]
        if isinstance(t, dict):
            return {k: leaves(v, dtype, grad) for k, v in t.items()}
        return t

    def flat(t):
        if torch.is_tensor(t):
            return [t]
        if isinstance(t, (list, tuple)):
            return [u for v in t for u in flat(v)]
        if isinstance(t, dict):
            return [u for v in t.values() for u in flat(v)]
        return []

    report, gen = [], torch.Generator(device="cuda").manual_seed(77)
    for name, mod in blocks:
        args, kwargs = rec.pop(name)
        cot, runs = None, []
        for amp in (False, True):
            mod.zero_grad(set_to_none=True)
            want_din = type(mod).__name__ != "StemBlock"          # (the image needs no gradient: stem1 has no data-gradient kernel)
            a = leaves(list(args), torch.bfloat16 if amp else torch.float32, want_din)
            kw = leaves(kwargs, torch.bfloat16 if amp else torch.float32, False)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = mod(*a, **kw)
            if cot is None:
                cot = torch.randn(out.shape, device=cuda, generator=gen)
            out.backward(cot.to(out.dtype))
            torch.cuda.synchronize()
            ins = [t.grad.detach().float() for t in flat(a) + flat(kw) if torch.is_tensor(t) and t.is_floating_point() and t.grad is not None]
            grads = {k: p.grad.detach().float().clone() for k, p in mod.named_parameters() if p.grad is not None}
            runs.append((out.detach().float(), ins, grads))
            mod.zero_grad(set_to_none=True)
        (oa, ia, ga), (ob, ib, gb) = runs
        assert set(ga) == set(gb) and len(ia) == len(ib) and (len(ia) >= 1 or type(mod).__name__ == "StemBlock")
        owners = dict(mod.named_modules())
        by = {"conv": [], "bn": [], "other": []}
        for k in ga:
            owner = owners[k.rsplit(".", 1)[0]] if "." in k else mod
            by["conv" if isinstance(owner, torch.nn.Conv2d) else "bn" if isinstance(owner, torch.nn.BatchNorm2d) else "other"].append(k)
        row = {"out": cos(ob, oa), "din": min([cos(x, y) for x, y in zip(ib, ia)] or [1.0])}
        for kind, keys in by.items():
            if keys:
                row[kind] = cos(torch.cat([gb[k].flatten() for k in keys]), torch.cat([ga[k].flatten() for k in keys]))
                row[kind + "_norm"] = (torch.cat([gb[k].flatten() for k in keys]).norm() / torch.cat([ga[k].flatten() for k in keys]).norm()).item()
        row["worst_conv"] = min([(cos(gb[k], ga[k]), k) for k in by["conv"] if ga[k].numel() >= 512] or [(1.0, "-")])
        report.append((name, type(mod).__name__, row))
        del runs, oa, ob, ia, ib, ga, gb
    if os.environ.get("DFINE_ANCHOR_PRINT") == "1":
        print()
        for name, kind, row in report:
            print(f"{name:28s} {kind:24s} " + " ".join(f"{k} {v:.5f}" if not isinstance(v, tuple) else f"{k} {v[0]:.4f} ({v[1]})" for k, v in row.items()))
    for name, kind, row in report:
        # measured (MI355X, profiles/r06_bf16_anchor.txt): backbone blocks out >= 0.99994, din >= 0.99018, conv >= 0.99069 (norm within
        # 0.3 %), BatchNorm affine >= 0.99258, LAB / other >= 0.99439, worst single conv weight 0.9851 (stem1); encoder blocks all >= 0.9998
        assert row["out"] >= 0.9995, (name, row)
        assert row["din"] >= 0.985, (name, row)
        if "conv" in row:
            assert row["conv"] >= 0.985 and abs(row["conv_norm"] - 1.0) <= 0.02, (name, row)
            assert row["worst_conv"][0] >= 0.97, (name, row)
        if "bn" in row:
            assert row["bn"] >= 0.985 and abs(row["bn_norm"] - 1.0) <= 0.03, (name, row)
        if "other" in row:
            assert row["other"] >= 0.985, (name, row)
