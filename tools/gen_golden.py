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
from tests import helpers  # noqa: E402  (before the reference import: that one re-binds `src`)
from ref_import import import_reference_dl  # noqa: E402

ref = import_reference_dl()
OUT = os.path.join(ROOT, "tests", "golden")


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


# ------------------------------------------------------------------ A18: post-processor
def gen_postprocess():
    out = {}
    for seed, kw in ((0, {}), (1, dict(B=3, Q=40, C=7))):
        logits, boxes, orig = helpers.make_postprocess_case(seed, **kw)
        C = logits.shape[-1]
        pp = ref.dl_export.DFINEPostProcessor(C, num_top_queries=300)
        o = {"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes)}
        for hh, ww in ((640, 640), (384, 512)):
            labels, bx, scores = pp(o, hh, ww)
            out[f"s{seed}/{hh}x{ww}/labels"] = labels.numpy()
            out[f"s{seed}/{hh}x{ww}/boxes"] = bx.numpy()
            out[f"s{seed}/{hh}x{ww}/scores"] = scores.numpy()
        # Trainer.preds_postprocess = process_boxes (map to the original image) + the same top-k (train.py:240-332;
        # train.py itself needs the whole training tool-chain to import, its box mapping lives in src/dl/utils.py)
        for keep_ratio in (False, True):
            pb = ref.dl_utils.process_boxes(torch.tensor(boxes), (640, 640), torch.tensor(orig), keep_ratio, "cpu")
            out[f"s{seed}/process_boxes/keep{int(keep_ratio)}"] = pb.numpy()
        out[f"s{seed}/orig_sizes"] = orig
    save("postprocess.npz", **out)
    # mask side of the evaluation hand-off: process_masks (network-size probabilities -> original frame, letterbox padding
    # removed) and cleanup_masks (pixels outside the box cleared), src/dl/utils.py:715-787
    out = {}
    g = torch.Generator().manual_seed(11)
    pm = torch.rand(2, 5, 20, 24, generator=g)
    orig = torch.tensor([[60, 100], [150, 75]])
    out["pred_masks"] = pm.numpy()
    out["orig_sizes"] = orig.numpy()
    for keep_ratio in (False, True):
        ml = ref.dl_utils.process_masks(pm, (80, 96), orig, keep_ratio)
        for b, m in enumerate(ml):
            out[f"process_masks/keep{int(keep_ratio)}/{b}"] = m.numpy().astype(np.float16)
            out[f"process_masks/keep{int(keep_ratio)}/{b}_bin"] = np.packbits((m >= 0.5).numpy(), axis=-1)
    boxes = torch.tensor([[10.0, 5.0, 60.0, 50.0], [0.0, 0.0, 100.0, 60.0], [30.5, 20.2, 31.0, 50.7], [75.0, 50.0, 99.0, 59.0], [5.0, 5.0, 5.0, 5.0]])
    mb = (ref.dl_utils.process_masks(pm[:1], (80, 96), orig[:1], False)[0] >= 0.5).to(torch.uint8)
    out["cleanup/boxes"] = boxes.numpy()
    out["cleanup/masks"] = np.packbits(ref.dl_utils.cleanup_masks(mb, boxes).numpy(), axis=-1)
    save("postprocess_masks.npz", **out)


# ------------------------------------------------------------------ A10 / A15: mask decoder, mask losses, mask costs
def _pool8(t):
    """[B,Q,H,W] -> [B,Q,8,8] area means: a compact fingerprint of the mask maps."""
    return torch.nn.functional.adaptive_avg_pool2d(t, 8)


def gen_mask_units():
    out = {}
    # MaskDecoder on small maps (ref dfine_decoder.py:316-370)
    torch.manual_seed(5)
    md = ref.decoder.MaskDecoder([64, 64, 64], out_ch=64)
    md.load_state_dict(helpers.seeded_state_dict(md.state_dict()))
    g = torch.Generator().manual_seed(6)
    feats = [torch.randn(2, 64, s, s, generator=g, requires_grad=True) for s in (20, 10, 5)]
    y = md(feats)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out["md/out"] = y.detach().numpy()
    out["md/grad_out"] = go.numpy()
    for i, f in enumerate(feats):
        out[f"md/feat{i}"] = f.detach().numpy()
        out[f"md/g_feat{i}"] = f.grad.numpy()
    out["md/g_up_conv"] = md.up_conv.weight.grad.numpy()
    out["md/g_lateral1"] = md.lateral[1].weight.grad.numpy()
    out["md/g_gn0_w"] = md.bn[0].weight.grad.numpy()

    # cropped BCE / Dice + target preparation (ref dfine_criterion.py:239-270,335-450,504-556)
    crit = ref.dfine.build_loss("n", 80, 0.0, True)
    targets = helpers.make_targets(2, 80, seed=9, mask_size=64)
    pm = torch.randn(2, 12, 16, 16, generator=g, requires_grad=True)
    indices = [(torch.tensor([1, 4, 7][:len(t["labels"])]), torch.arange(len(t["labels"]))[:3]) for t in targets]
    losses = crit.loss_masks({"pred_masks": pm}, targets, indices, 1.0)
    (losses["loss_mask_bce"] + 2 * losses["loss_mask_dice"]).backward()
    out["loss/pred_masks"] = pm.detach().numpy()
    out["loss/bce"] = losses["loss_mask_bce"].detach().numpy()
    out["loss/dice"] = losses["loss_mask_dice"].detach().numpy()
    out["loss/g_pred_masks"] = pm.grad.numpy()
    for b, (i, j) in enumerate(indices):
        out[f"loss/rows{b}"] = i.numpy()
        out[f"loss/cols{b}"] = j.numpy()

    # matcher with mask costs (ref matcher.py:19-71,175-237)
    matcher = ref.matcher.HungarianMatcher(**ref.configs.models["n"]["matcher"])
    logits, boxes, _ = helpers.make_matcher_case(3, B=2, Q=12, C=80, sizes=(3, 3))
    captured = []
    orig = ref.matcher.linear_sum_assignment

    def spy(c):
        captured.append(np.array(c, copy=True))
        return orig(c)

    ref.matcher.linear_sum_assignment = spy
    res = matcher({"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes),
                   "pred_masks": pm.detach()}, targets)
    ref.matcher.linear_sum_assignment = orig
    for b, ((i, j), c) in enumerate(zip(res["indices"], captured)):
        out[f"match/rows{b}"] = i.numpy()
        out[f"match/cols{b}"] = j.numpy()
        out[f"match/cost{b}"] = c.astype(np.float32)
    save("mask_units.npz", **out)


def gen_mask_model():
    """D-FINE-n + segmentation head, 320x320, bs 2 (the code path of BASELINE config #5 at a size the CPU reference
    finishes in seconds): eval boxes/logits/mask fingerprints, all train losses incl. the mask terms, mask-path gradients."""
    torch.manual_seed(0)
    model = ref.dfine.build_model("n", 80, True, "cpu", img_size=[320, 320])
    model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
    x = helpers.make_images(2, 320)
    out = {}
    model.eval()
    with torch.no_grad():
        o = model(x)
    out["eval/pred_logits"] = o["pred_logits"].numpy()
    out["eval/pred_boxes"] = o["pred_boxes"].numpy()
    out["eval/pred_masks_pool8"] = _pool8(o["pred_masks"]).numpy()
    targets = helpers.make_targets(2, 80, mask_size=320)
    crit = ref.dfine.build_loss("n", 80, 0.0, True)
    model.train()
    torch.manual_seed(11)
    o = model(x, targets)
    losses = crit(o, targets)
    sum(losses.values()).backward()
    for k, v in losses.items():
        out[f"train/loss/{k}"] = v.detach().numpy()
    for k in ("decoder.mask_decoder.up_conv.weight", "decoder.mask_decoder.lateral.0.weight",
              "decoder.mask_head.layers.2.weight", "encoder.input_proj.0.conv.weight",
              "decoder.dec_bbox_head.1.layers.2.weight"):
        out[f"train/grad/{k}"] = dict(model.named_parameters())[k].grad.numpy()
    save("model_n320_mask.npz", **out)


def gen_backbone_encoder():
    """HGNetv2 + HybridEncoder of D-FINE-m at 320x320 (no discrete selections inside): features and the parameter
    gradients of sum(feature * fixed cotangent) - the fp32 anchor for the bf16 MFMA path (convs, stem, BN, AIFI)."""
    torch.manual_seed(0)
    model = ref.dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
    model.train()
    x = helpers.make_images(2, 320)
    feats = model.encoder(model.backbone(x))
    out = {}
    loss = 0
    for i, f in enumerate(feats):
        go = helpers.make_cotangent(f.shape, 50 + i)
        out[f"feat{i}"] = f.detach().numpy()[:1].astype(np.float16)   # image 0; 1e-3 relative is ample for these checks
        loss = loss + (f * go).sum()
    loss.backward()
    params = dict(model.named_parameters())
    for k in helpers.BACKBONE_ENCODER_GRAD_KEYS:
        gk = helpers.compact_rows(params[k].grad.numpy())
        out[f"grad/{k}"] = (gk / np.abs(gk).max()).astype(np.float16)  # unit-scaled (fp16 would flush the small ones)
        out[f"gscale/{k}"] = np.float32(np.abs(gk).max())
    save("backbone_encoder_m320.npz", **out)


def gen_deploy():
    """`model.deploy()` of the reference (dfine.py:43-48: eval + convert_to_deploy of every module - conv-BN folding
    hybrid_encoder.py:47-79, RepVGG re-parameterisation :123-156, decoder pruning dfine_decoder.py:422-427,698-707) on the
    seeded D-FINE-n 320x320 model: eval outputs after deploy, the deployed module inventory, and - for D-FINE-m, whose
    encoder carries every deployable unit type at MFMA-eligible channel counts - the encoder features after deploy."""
    out = {}
    for size, tag in (("n", "n320"), ("m", "m320")):
        torch.manual_seed(0)
        model = ref.dfine.build_model(size, 80, False, "cpu", img_size=[320, 320])
        model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
        x = helpers.make_images(2, 320)
        model.eval()
        with torch.no_grad():
            before = model(x)
        model.deploy()
        with torch.no_grad():
            feats = model.encoder(model.backbone(x))
            o = model(x)
        out[f"{tag}/pred_logits"] = o["pred_logits"].numpy()
        out[f"{tag}/pred_boxes"] = o["pred_boxes"].numpy()
        out[f"{tag}/before_logits"] = before["pred_logits"].numpy()          # eval outputs of the un-deployed model
        out[f"{tag}/before_boxes"] = before["pred_boxes"].numpy()
        if size == "m":
            for i, f in enumerate(feats):
                out[f"{tag}/feat{i}"] = f.numpy()[:1, ::2].astype(np.float16)   # image 0, every other channel
        out[f"{tag}/state_keys"] = np.array(sorted(model.state_dict().keys()))
        sd = model.state_dict()
        for k in ("encoder.fpn_blocks.0.cv2.0.bottlenecks.0.conv.weight", "encoder.fpn_blocks.0.cv2.0.bottlenecks.0.conv.bias",
                  "encoder.input_proj.0.conv_bn_fused.weight" if "encoder.input_proj.0.conv_bn_fused.weight" in sd else
                  "encoder.lateral_convs.0.conv_bn_fused.weight", "encoder.lateral_convs.0.conv_bn_fused.bias"):
            out[f"{tag}/w/{k}"] = helpers.compact_rows(sd[k].numpy())
    save("deploy.npz", **out)


def gen_validator():
    """Box metrics of the reference's `Validator` (src/dl/validator.py:295-451: greedy IoU matching, per-class TP / FP / FN /
    IoU lists, confusion matrix) on seeded detection lists; `compute_maps=False` (torchmetrics / faster_coco_eval are not in
    this container) and torchvision's `box_iou` supplied as the standard pairwise IoU."""
    import importlib
    import torchvision

    def box_iou(a, b):
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (area_a[:, None] + area_b[None] - inter)

    torchvision.ops.box_iou = box_iou
    V = importlib.import_module("src.dl.validator")
    assert V.__file__.startswith("/root/reference/")
    out = {}
    for seed in (1, 2, 3):
        gt, preds = helpers.make_validator_case(seed)
        for thr in (0.5, 0.75):
            import copy
            v = V.Validator(copy.deepcopy(gt), copy.deepcopy(preds), {i: f"c{i}" for i in range(5)}, conf_thresh=0.5,
                            iou_thresh=thr, compute_maps=False)
            m = v.compute_metrics(extended=True)
            k = f"s{seed}_t{int(thr * 100)}"
            for name in ("f1", "precision", "recall", "iou", "TPs", "FPs", "FNs"):
                out[f"{k}/{name}"] = np.float64(m[name])
            ext = m["extended_metrics"]
            out[f"{k}/ext_keys"] = np.array(sorted(ext))
            out[f"{k}/ext_vals"] = np.array([float(ext[x]) for x in sorted(ext)], dtype=np.float64)
            out[f"{k}/conf_matrix"] = v.conf_matrix
            out[f"{k}/classes"] = np.array(sorted(v.class_to_idx))
    save("validator.npz", **out)
    # instance masks: the matching runs on the pairwise mask IoU (validator.py:453-568); dense uint8 masks, float mask_probs
    # (binarised with > conf_thresh) and predictions of another resolution (bilinear resize + > 0.5)
    out = {}
    for name, kw in (("dense", {}), ("probs", {"probs": True}), ("resized", {"pred_hw": (48, 64)}), ("resized_probs", {"pred_hw": (60, 80), "probs": True})):
        for seed in (1, 2):
            gt, preds = helpers.make_validator_mask_case(seed, **kw)
            import copy
            v = V.Validator(copy.deepcopy(gt), copy.deepcopy(preds), {i: f"c{i}" for i in range(5)}, conf_thresh=0.5,
                            iou_thresh=0.5, compute_maps=False)
            assert v.use_masks
            m = v.compute_metrics(extended=True)
            k = f"{name}_s{seed}"
            for nm in ("f1", "precision", "recall", "iou", "TPs", "FPs", "FNs"):
                out[f"{k}/{nm}"] = np.float64(m[nm])
            ext = m["extended_metrics"]
            out[f"{k}/ext_keys"] = np.array(sorted(ext))
            out[f"{k}/ext_vals"] = np.array([float(ext[x]) for x in sorted(ext)], dtype=np.float64)
            out[f"{k}/conf_matrix"] = v.conf_matrix
            # the pairwise IoU matrices themselves (bit-exact target of the packed-mask kernels), dense same-size case only
            if name == "dense":
                for i, (p, g) in enumerate(zip(preds, gt)):
                    out[f"{k}/iou{i}"] = v._pairwise_mask_iou(p["masks"], g["masks"]).numpy().astype(np.float32) \
                        if len(p["labels"]) and len(g["labels"]) else np.zeros((len(p["labels"]), len(g["labels"])), np.float32)
    save("validator_masks.npz", **out)


# ------------------------------------------------------------------ (f3) augmentation box path: random_affine / box_candidates / mosaic coordinates
def gen_data_device():
    """The box side of the reference's mosaic + affine augmentation (src/dl/utils.py:283-414), image warps excluded (cv2 is not
    in the container): `random_affine` itself with its matrix draw pinned (get_transform_matrix patched to return the given
    matrix) and cv2.warpAffine inert, `box_candidates` and `get_mosaic_coordinate` called directly."""
    U = ref.dl_utils
    rs = np.random.RandomState(7)
    out = {}
    n_cases = 6
    for c in range(n_cases):
        tw, th = [(640, 640), (320, 320), (640, 480)][c % 3]
        a, sc = np.radians(rs.uniform(-10, 10)), rs.uniform(0.5, 1.5)
        if c == 0:
            a, sc = 0.0, 0.5
        shx, shy = np.tan(np.radians(rs.uniform(-2, 2))), np.tan(np.radians(rs.uniform(-2, 2)))
        tx, ty = rs.uniform(0.4, 0.6) * tw, rs.uniform(0.4, 0.6) * th
        R = np.array([[sc * np.cos(a), sc * np.sin(a), 0], [-sc * np.sin(a), sc * np.cos(a), 0], [0, 0, 1.0]])
        S = np.array([[1, shx, 0], [shy, 1, 0], [0, 0, 1.0]])
        T = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1.0]])
        C = np.array([[1, 0, -tw], [0, 1, -th], [0, 0, 1.0]])           # canvas of 2 x target: centre (tw, th)
        M = T @ S @ R @ C
        n = 120
        xy = rs.uniform(0, 2 * np.array([tw, th]) - 10, (n, 2))
        wh = rs.uniform(1, 0.6 * np.array([tw, th]), (n, 2))
        x2y2 = np.minimum(xy + wh, 2 * np.array([tw, th]))
        targets = np.concatenate([rs.randint(0, 5, (n, 1)), xy, x2y2], 1).astype(np.float32)
        orig = U.get_transform_matrix
        U.get_transform_matrix = lambda *a_, **k_: (M, sc)
        try:
            img = np.zeros((2 * th, 2 * tw, 3), np.uint8)
            _, t_aff, _ = U.random_affine(img, targets.copy(), None, (tw, th), 10.0, 0.1, (0.5, 1.5), 2.0)
        finally:
            U.get_transform_matrix = orig
        out[f"affine{c}/M"] = M
        out[f"affine{c}/scale"] = np.float64(sc)
        out[f"affine{c}/target_size"] = np.array([tw, th])
        out[f"affine{c}/targets_in"] = targets
        out[f"affine{c}/targets_out"] = np.asarray(t_aff, dtype=np.float32)
    out["n_affine"] = np.int64(n_cases)
    b1 = rs.uniform(0, 600, (4, 300)).astype(np.float32)
    b1[2:] += b1[:2]
    b2 = b1 * rs.uniform(0.0, 1.3, (1, 300)).astype(np.float32) + rs.uniform(-3, 3, (4, 300)).astype(np.float32)
    out["cand/box1"], out["cand/box2"] = b1, b2
    out["cand/keep"] = U.box_candidates(b1, b2, area_thr=0.1)
    rows = []
    for idx in range(4):
        for xc, yc, w, h in ((700, 500, 640, 480), (400, 390, 640, 480), (960, 960, 300, 700), (330, 610, 1000, 200), (640, 640, 640, 640)):
            (x1, y1, x2, y2), small = U.get_mosaic_coordinate(None, idx, xc, yc, w, h, 640, 640)
            rows.append([idx, xc, yc, w, h, x1, y1, x2, y2, *small])
    out["mosaic/rows"] = np.asarray(rows, dtype=np.int64)
    save("data_device.npz", **out)


# ------------------------------------------------------------------ A16: parameter-group membership of build_optimizer, all sizes
def gen_param_groups():
    out = {}
    for size, mask in (("n", False), ("s", False), ("m", False), ("l", False), ("x", False), ("x", True)):
        model = ref.dfine.build_model(size, 80, mask, "cpu", img_size=[640, 640])
        opt = ref.dfine.build_optimizer(model, lr=1.5e-4, backbone_lr=2e-5, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=1.5e-4)
        gid = {}
        for g, grp in enumerate(opt.param_groups):
            for p in grp["params"]:
                gid[id(p)] = g
        names = [n for n, _ in model.named_parameters()]
        groups = [gid[id(p)] for _, p in model.named_parameters()]
        tag = size + ("_mask" if mask else "")
        out[f"{tag}/names"] = np.array(names)
        out[f"{tag}/group"] = np.asarray(groups, dtype=np.int8)
        out[f"{tag}/requires_grad"] = np.asarray([p.requires_grad for _, p in model.named_parameters()])
        out[f"{tag}/lr"] = np.asarray([grp["lr"] for grp in opt.param_groups], dtype=np.float64)
        out[f"{tag}/weight_decay"] = np.asarray([grp["weight_decay"] for grp in opt.param_groups], dtype=np.float64)
    save("param_groups.npz", **out)


# ------------------------------------------------------------------ the train loop: lr schedule x clip x AdamW groups x EMA across iterations
def gen_train_trace():
    """Three iterations of the reference's loop (src/dl/train.py:512-535 optimizer_step, :550-586 the step, :52-73 ModelEMA,
    :203-221 OneCycleLR) RESTATED here around the imported build_model / build_loss / build_optimizer - `src.dl.train`
    itself cannot be imported (hydra, wandb, cv2 ... are not in the container).  fp32, no AMP, D-FINE-n 320 x 320, bs 2."""
    import math
    from copy import deepcopy
    base_lr, backbone_lr, iters = 8e-4, 4e-4, 3
    torch.manual_seed(0)
    model = ref.dfine.build_model("n", 80, False, "cpu", img_size=[320, 320])
    model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
    crit = ref.dfine.build_loss("n", 80, 0.0, False)
    opt = ref.dfine.build_optimizer(model, lr=base_lr, backbone_lr=backbone_lr, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=base_lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=base_lr * 2, epochs=1, steps_per_epoch=8, pct_start=0.1, cycle_momentum=False)
    ema = deepcopy(model).eval()
    for p in ema.parameters():
        p.requires_grad_(False)
    ema_momentum = 0.9998
    watch = ("backbone.stem.stem1.conv.weight", "encoder.input_proj.0.conv.weight", "decoder.enc_score_head.bias",
             "decoder.dec_bbox_head.1.layers.2.weight", "backbone.stages.0.blocks.0.layers.0.bn.weight")
    ema_watch = watch + ("backbone.stem.stem1.bn.running_mean", "backbone.stem.stem1.bn.running_var")
    out = {"iters": np.int64(iters), "base_lr": np.float64(base_lr), "backbone_lr": np.float64(backbone_lr)}
    model.train()
    crit.train()
    for it in range(iters):
        x = helpers.make_images(2, 320, seed=500 + it)
        targets = helpers.make_targets(2, 80)
        out[f"it{it}/lr"] = np.asarray([g["lr"] for g in opt.param_groups], dtype=np.float64)
        torch.manual_seed(11 + it)                      # CDN noise of this iteration (CPU generator)
        o = model(x, targets=targets)
        loss_dict = crit(o, targets)
        loss = sum(loss_dict.values()) / 1
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        sched.step()
        opt.zero_grad()
        momentum = ema_momentum * (1 - math.exp(-(it + 1) / 2000))
        student = model.state_dict()
        with torch.no_grad():
            for name, param in ema.state_dict().items():
                if param.dtype.is_floating_point:
                    param *= momentum
                    param += (1.0 - momentum) * student[name].detach()
        out[f"it{it}/loss"] = np.float64(loss.item())
        out[f"it{it}/grad_norm"] = np.float64(float(norm))
        out[f"it{it}/ema_momentum"] = np.float64(momentum)
        for k, v in loss_dict.items():
            out[f"it{it}/losses/{k}"] = np.float64(v.item())
    sd, esd = model.state_dict(), ema.state_dict()
    for k in watch:
        out[f"final/{k}"] = sd[k].numpy()
    for k in ema_watch:
        out[f"final_ema/{k}"] = esd[k].numpy()
    out["final/num_batches_tracked"] = np.int64(sd["backbone.stem.stem1.bn.num_batches_tracked"].item())
    save("train_trace.npz", **out)


GENERATORS = {
    "lsap": gen_lsap, "msda": gen_msda, "matcher": gen_matcher, "criterion": gen_criterion,
    "model_n320": lambda: gen_model("n", 320, 2, "model_n320.npz"),
    "model_m640_eval": lambda: gen_model("m", 640, 1, "model_m640_eval.npz", train=False),
    "model_m640_eval_b3": lambda: gen_model("m", 640, 3, "model_m640_eval_b3.npz", train=False),
    "model_s320": lambda: gen_model("s", 320, 2, "model_s320.npz"),
    "backbone_encoder": gen_backbone_encoder, "postprocess": gen_postprocess, "mask_units": gen_mask_units, "model_n320_mask": gen_mask_model,
    "validator": gen_validator, "deploy": gen_deploy,
    "data_device": gen_data_device, "param_groups": gen_param_groups, "train_trace": gen_train_trace,
}

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT, help="output directory (default tests/golden)")
    ap.add_argument("only", nargs="*", help=f"subset of {sorted(GENERATORS)}")
    a = ap.parse_args()
    OUT = a.out
    os.makedirs(OUT, exist_ok=True)
    for name in (a.only or GENERATORS):
        GENERATORS[name]()
