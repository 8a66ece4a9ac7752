"""Generates tests/golden/*.npz by running the REFERENCE (/root/reference, imported here only) on
the deterministic inputs of tests/helpers.py.  Run in the build container:

    python tools/gen_golden.py

The fixtures hold inputs/outputs only (no reference source).  The reference ships no tests or
golden vectors of its own for this path, so these files are what pins the
oracle (oracle/np_ref.py, oracle/lsap.c, oracle/torch_backend.py) and, through it, the HIP kernels.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from tests import helpers  # noqa: E402

ref = import_reference()
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ------------------------------------------------------------------ A12: SciPy LSAP
def gen_lsap():
    from scipy.optimize import linear_sum_assignment
    import scipy
    out = {"scipy_version": np.array(scipy.__version__)}
    for name, c in helpers.lsap_cases().items():
        r, k = linear_sum_assignment(c)
        out[name + "/cost"] = c
        out[name + "/rows"] = r.astype(np.int64)
        out[name + "/cols"] = k.astype(np.int64)
    save("lsap.npz", **out)


# ------------------------------------------------------------------ A7: deformable gather
def gen_msda():
    out = {}
    for seed, kw in ((0, {}), (1, dict(B=1, Lq=5, H=2, D=16, shapes=((5, 7), (3, 2)), points=(2, 4)))):
        value, loc, w, go, shapes, points = helpers.make_msda_case(seed, **kw)
        v = torch.tensor(value, requires_grad=True)
        lc = torch.tensor(loc, requires_grad=True)
        ww = torch.tensor(w, requires_grad=True)
        # the reference takes per-level [B, H, D, h*w] lists
        split = [h * wd for h, wd in shapes]
        vlist = v.permute(0, 2, 3, 1).split(split, dim=-1)
        o = ref.arch_utils.deformable_attention_core_func_v2(vlist, [list(s) for s in shapes], lc, ww, list(points))
        o.backward(torch.tensor(go))
        out[f"s{seed}/out"] = o.detach().numpy()
        out[f"s{seed}/g_value"] = v.grad.numpy()
        out[f"s{seed}/g_loc"] = lc.grad.numpy()
        out[f"s{seed}/g_weight"] = ww.grad.numpy()
    # full module (softmax + location arithmetic + gather) with 4-d reference boxes
    torch.manual_seed(3)
    mod = ref.decoder.MSDeformableAttention(embed_dim=32, num_heads=8, num_levels=3, num_points=[3, 6, 3])
    torch.nn.init.normal_(mod.sampling_offsets.weight, std=0.3)
    torch.nn.init.normal_(mod.attention_weights.weight, std=0.5)
    shapes = [[8, 8], [4, 4], [2, 2]]
    B, Lq = 2, 9
    g = torch.Generator().manual_seed(4)
    query = torch.randn(B, Lq, 32, generator=g)
    refb = torch.cat([torch.rand(B, Lq, 2, generator=g), torch.rand(B, Lq, 2, generator=g) * 0.5], -1)
    value = torch.randn(B, 84, 8, 4, generator=g, requires_grad=True)
    cap = {}
    h1 = mod.sampling_offsets.register_forward_hook(lambda m, i, o: cap.__setitem__("off", o))
    h2 = mod.attention_weights.register_forward_hook(lambda m, i, o: cap.__setitem__("log", o))
    vlist = value.permute(0, 2, 3, 1).split([64, 16, 4], dim=-1)
    o = mod(query, refb.unsqueeze(2), vlist, shapes)
    h1.remove(); h2.remove()
    cap["off"].retain_grad(); cap["log"].retain_grad()
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    out.update({"mod/value": value.detach().numpy(), "mod/ref": refb.numpy(),
                "mod/offsets": cap["off"].detach().reshape(B, Lq, 8, 12, 2).numpy(),
                "mod/logits": cap["log"].detach().reshape(B, Lq, 8, 12).numpy(),
                "mod/grad_out": go.numpy(), "mod/out": o.detach().numpy(),
                "mod/g_value": value.grad.numpy(),
                "mod/g_offsets": cap["off"].grad.reshape(B, Lq, 8, 12, 2).numpy(),
                "mod/g_logits": cap["log"].grad.reshape(B, Lq, 8, 12).numpy()})
    save("msda.npz", **out)


# ------------------------------------------------------------------ A11: matcher
def gen_matcher():
    out = {}
    matcher = ref.matcher.HungarianMatcher(**ref.configs.models["m"]["matcher"])
    for seed, kw in ((0, {}), (1, dict(B=2, Q=300, C=80, sizes=(7, 23))), (2, dict(B=2, Q=6, C=4, sizes=(9, 2)))):
        logits, boxes, targets = helpers.make_matcher_case(seed, **kw)
        captured = []
        orig = ref.matcher.linear_sum_assignment

        def spy(c):
            captured.append(np.array(c, copy=True))
            return orig(c)

        ref.matcher.linear_sum_assignment = spy
        res = matcher({"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes)}, targets)
        ref.matcher.linear_sum_assignment = orig
        for b, ((i, j), c) in enumerate(zip(res["indices"], captured)):
            out[f"s{seed}/rows{b}"] = i.numpy()
            out[f"s{seed}/cols{b}"] = j.numpy()
            out[f"s{seed}/cost{b}"] = c.astype(np.float32)
    save("matcher.npz", **out)


# ------------------------------------------------------------------ A13/A14: criterion
def gen_criterion():
    out = {}
    crit = ref.dfine.build_loss("s", 6, 0.0, False)
    crit.num_classes = 6
    for seed in (0, 1):
        outputs = helpers.make_criterion_outputs(seed)
        targets, meta = helpers.criterion_targets_and_meta()
        outputs["dn_meta"] = meta
        losses = crit(outputs, targets)
        total = sum(losses.values())
        total.backward()
        for k, v in losses.items():
            out[f"s{seed}/loss/{k}"] = v.detach().numpy()
        out[f"s{seed}/grad/pred_logits"] = outputs["pred_logits"].grad.numpy()
        out[f"s{seed}/grad/pred_boxes"] = outputs["pred_boxes"].grad.numpy()
        out[f"s{seed}/grad/pred_corners"] = outputs["pred_corners"].grad.numpy()
        out[f"s{seed}/grad/aux0_corners"] = outputs["aux_outputs"][0]["pred_corners"].grad.numpy()
        out[f"s{seed}/grad/dn0_logits"] = outputs["dn_outputs"][0]["pred_logits"].grad.numpy()
        out[f"s{seed}/grad/enc_boxes"] = outputs["enc_aux_outputs"][0]["pred_boxes"].grad.numpy()
    save("criterion.npz", **out)


# ------------------------------------------------------------------ full model
def gen_model(size, img, batch, name, train=True):
    torch.manual_seed(0)
    model = ref.dfine.build_model(size, 80, False, "cpu", img_size=[img, img])
    model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
    x = helpers.make_images(batch, img)
    out = {}
    model.eval()
    with torch.no_grad():
        o = model(x)
    out["eval/pred_logits"] = o["pred_logits"].numpy()
    out["eval/pred_boxes"] = o["pred_boxes"].numpy()
    if train:
        targets = helpers.make_targets(batch, 80)
        crit = ref.dfine.build_loss(size, 80, 0.0, False)
        model.train()
        torch.manual_seed(11)  # CDN noise (CPU generator; this build draws in the same order)
        o = model(x, targets)
        losses = crit(o, targets)
        sum(losses.values()).backward()
        for k, v in losses.items():
            out[f"train/loss/{k}"] = v.detach().numpy()
        out["train/pred_logits"] = o["pred_logits"].detach().numpy()
        out["train/pred_boxes"] = o["pred_boxes"].detach().numpy()
        for k in ("backbone.stem.stem1.conv.weight", "encoder.input_proj.0.conv.weight",
                  "decoder.decoder.layers.0.cross_attn.sampling_offsets.weight",
                  "decoder.enc_score_head.weight", "decoder.dec_bbox_head.1.layers.2.weight"):
            out[f"train/grad/{k}"] = dict(model.named_parameters())[k].grad.numpy()
    save(name, **out)


if __name__ == "__main__":
    gen_lsap()
    gen_msda()
    gen_matcher()
    gen_criterion()
    gen_model("n", 320, 2, "model_n320.npz")
    gen_model("m", 640, 1, "model_m640_eval.npz", train=False)
