"""Deterministic inputs shared by tools/gen_golden.py (run once against the reference in the build
container) and the tests (run anywhere).  Nothing here touches /root/reference."""
import zlib

import numpy as np
import torch

GOLDEN_DIR = __file__.rsplit("/", 1)[0] + "/golden"


def seeded_state_dict(state_dict):
    """Pseudo-random weights derived from the KEY NAMES, so the reference model (in the generator)
    and this build's model (in the tests) get identical weights without shipping them.
    Structured / geometric tensors (anchors, masks, FDR constants, the deformable-offset ray
    bias) keep their constructor values - both code bases compute them by the same formula."""
    out = {}
    for k, v in state_dict.items():
        keep = (not v.dtype.is_floating_point or k.endswith(("anchors", "num_points_scale"))
                or k in ("decoder.up", "decoder.reg_scale") or k.endswith("sampling_offsets.bias"))
        if keep:
            out[k] = v.clone()
            continue
        rng = np.random.default_rng(zlib.crc32(k.encode()))
        shape = tuple(v.shape)
        is_norm = any(s in k for s in (".bn.", ".norm.", "norm1.", "norm2.", "norm3.", ".norm"))
        if k.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            a = rng.normal(0, 0.1, shape)
        elif k.endswith("lab.scale"):
            a = rng.uniform(0.9, 1.1, shape)
        elif k.endswith("lab.bias"):
            a = rng.normal(0, 0.02, shape)
        elif is_norm and k.endswith("weight"):
            a = rng.uniform(0.8, 1.2, shape)
        elif k.endswith("bias"):
            a = rng.normal(0, 0.05, shape)
        elif "class_embed" in k:
            a = rng.normal(0, 1.0, shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            a = rng.normal(0, 1.0 / np.sqrt(max(fan_in, 1)), shape)
        out[k] = torch.from_numpy(a.astype(np.float32)).reshape(v.shape)
    return out


BACKBONE_ENCODER_GRAD_KEYS = (
    "backbone.stem.stem1.conv.weight", "backbone.stem.stem3.conv.weight", "backbone.stages.0.blocks.0.layers.0.conv.weight",
    "backbone.stages.1.blocks.0.aggregation.0.conv.weight", "backbone.stages.2.blocks.0.layers.1.conv2.conv.weight",
    "backbone.stages.2.blocks.1.layers.0.conv1.conv.weight", "backbone.stages.3.blocks.0.aggregation.1.conv.weight",
    "backbone.stages.2.blocks.0.layers.0.conv1.bn.weight", "encoder.input_proj.0.conv.weight",
    "encoder.encoder.0.layers.0.self_attn.in_proj_weight", "encoder.encoder.0.layers.0.linear1.weight",
    "encoder.fpn_blocks.0.cv1.conv.weight", "encoder.fpn_blocks.0.cv2.0.bottlenecks.0.conv1.conv.weight",
    "encoder.pan_blocks.0.cv4.conv.weight")


def compact_rows(t, rows=64):
    """First `rows` rows of a tensor / array (golden fixtures keep a slice of the big weight gradients)."""
    return t[:rows] if t.shape[0] > rows else t


def make_cotangent(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).normal(0, 1, tuple(shape)).astype(np.float32))


def make_images(batch, size, seed=123):
    return torch.from_numpy(np.random.default_rng(seed).random((batch, 3, size, size), np.float32))


def box_masks(boxes, size):
    """Instance masks of the segmentation config: ellipses inscribed in the GT boxes,
    uint8 [T, size, size] (an ellipse, not the filled rectangle, so that the box-cropped losses see both classes)."""
    ys = (np.arange(size, dtype=np.float32) + 0.5)[None, :, None] / size
    xs = (np.arange(size, dtype=np.float32) + 0.5)[None, None, :] / size
    b = np.asarray(boxes, dtype=np.float32)
    cx, cy, w, h = (b[:, i][:, None, None] for i in range(4))
    return ((((xs - cx) / (w / 2)) ** 2 + ((ys - cy) / (h / 2)) ** 2) <= 1.0).astype(np.uint8)


def make_targets(batch, num_classes, seed=7, max_t=6, min_t=1, device="cpu", mask_size=None):
    """COCO-shaped synthetic labels: boxes stay inside the image (cx,cy in [.2,.8], w,h in [.05,.35]).
    `mask_size`: also emit `masks` u8 [T, mask_size, mask_size] (segmentation task)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batch):
        n = int(rng.integers(min_t, max_t + 1))
        cxcy = rng.uniform(0.2, 0.8, (n, 2))
        wh = rng.uniform(0.05, 0.35, (n, 2))
        boxes = np.concatenate([cxcy, wh], 1).astype(np.float32)
        t = {
            "labels": torch.from_numpy(rng.integers(0, num_classes, n)).long().to(device),
            "boxes": torch.from_numpy(boxes).to(device),
        }
        if mask_size is not None:
            t["masks"] = torch.from_numpy(box_masks(boxes, mask_size)).to(device)
        out.append(t)
    return out


def make_postprocess_case(seed, B=2, Q=300, C=80):
    """Decoder outputs for the post-processor goldens: logits with exact ties (duplicated queries, a constant row)
    and boxes that poke outside the image (exercises the floor/ceil clamps)."""
    rng = np.random.default_rng(seed)
    logits = rng.normal(-2.0, 2.0, (B, Q, C)).astype(np.float32)
    logits[0, 5] = logits[0, 3]                      # tie between two queries
    logits[B - 1, 7, :] = 0.25                       # tie inside one query
    boxes = np.concatenate([rng.uniform(0.0, 1.0, (B, Q, 2)), rng.uniform(0.01, 0.6, (B, Q, 2))], -1).astype(np.float32)
    orig = np.array([[480, 640], [1080, 1920], [333, 500], [640, 640]][:B], dtype=np.int64)
    return logits, boxes, orig


def make_msda_case(seed, B=2, Lq=16, H=8, D=4, shapes=((8, 8), (4, 4), (2, 2)), points=(3, 6, 3)):
    """Random value / sampling locations / weights incl. out-of-range and exactly-on-edge points."""
    rng = np.random.default_rng(seed)
    L = sum(h * w for h, w in shapes)
    P = sum(points)
    value = rng.normal(0, 1, (B, L, H, D)).astype(np.float32)
    loc = rng.uniform(-0.15, 1.15, (B, Lq, H, P, 2)).astype(np.float32)
    # exactly on pixel centres / borders / far outside
    loc[0, 0, 0, :4] = [[0.0, 0.0], [1.0, 1.0], [0.5, 0.5], [0.0625, 0.9375]]
    loc[0, 1, 1, :3] = [[-3.0, 0.5], [0.5, 7.0], [1e4, -1e4]]
    w = rng.uniform(0, 1, (B, Lq, H, P)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    grad_out = rng.normal(0, 1, (B, Lq, H * D)).astype(np.float32)
    return value, loc, w, grad_out, tuple(shapes), tuple(points)


def make_matcher_case(seed, B=3, Q=40, C=7, sizes=(5, 0, 9)):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2, (B, Q, C)).astype(np.float32)
    boxes = np.concatenate([rng.uniform(0.2, 0.8, (B, Q, 2)), rng.uniform(0.05, 0.4, (B, Q, 2))], -1).astype(np.float32)
    targets = []
    for n in sizes:
        targets.append({
            "labels": torch.from_numpy(rng.integers(0, C, n)).long(),
            "boxes": torch.from_numpy(np.concatenate([rng.uniform(0.2, 0.8, (n, 2)), rng.uniform(0.05, 0.4, (n, 2))], -1).astype(np.float32)),
        })
    return logits, boxes, targets


def lsap_cases():
    """Assignment problems pinned against SciPy in tests/golden/lsap.npz (name -> cost matrix)."""
    rng = np.random.default_rng(2024)
    cases = {}
    for t in (1, 7, 50, 100, 300, 350):
        cases[f"rand_300x{t}"] = rng.random((300, t)).astype(np.float32)
    cases["zeros_6x3"] = np.zeros((6, 3), np.float32)
    cases["dup_rows"] = np.array([[1, 1, 1], [1, 1, 1], [0, 0, 0], [1, 1, 1], [0, 0, 0]], np.float32)
    cases["all_equal_300x20"] = np.full((300, 20), 0.25, np.float32)
    cases["small_ints_40x12"] = rng.integers(0, 3, (40, 12)).astype(np.float32)
    cases["small_ints_12x40"] = rng.integers(0, 3, (12, 40)).astype(np.float32)
    cases["coarse_300x30"] = np.round(rng.random((300, 30)), 1).astype(np.float32)
    dup = rng.random((300, 10)).astype(np.float32)
    dup[:, 5] = dup[:, 2]
    dup[100] = dup[7]
    cases["dup_cols_rows_300x10"] = dup
    nanc = rng.random((50, 6)).astype(np.float32)
    nanc[3, 2] = np.nan
    nanc[10, :] = np.nan
    cases["nan_to_one_50x6"] = np.nan_to_num(nanc, nan=1.0)
    return cases


def make_criterion_outputs(seed, B=2, Q=24, C=6, L=3, dn=8, reg_max=32, device="cpu", requires_grad=True):
    """A synthetic decoder output dict with the train-time structure of DFINETransformer.forward
    (L decoder layers -> L-1 aux heads, pre, enc, dn heads) on tiny shapes."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale=1.0):
        t = (torch.randn(*shape, generator=g) * scale).to(device)
        return t.requires_grad_(requires_grad)

    def boxes(*lead):
        cxcy = torch.rand(*lead, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(*lead, 2, generator=g) * 0.3 + 0.05
        return torch.cat([cxcy, wh], -1).to(device).requires_grad_(requires_grad)

    nb = 4 * (reg_max + 1)
    ref = boxes(B, Q).detach()
    dn_ref = boxes(B, dn).detach()
    layers = [{"pred_logits": rnd(B, Q, C), "pred_boxes": boxes(B, Q), "pred_corners": rnd(B, Q, nb),
               "ref_points": ref} for _ in range(L)]
    dn_layers = [{"pred_logits": rnd(B, dn, C), "pred_boxes": boxes(B, dn), "pred_corners": rnd(B, dn, nb),
                  "ref_points": dn_ref} for _ in range(L)]
    up = torch.tensor([0.5], device=device)
    reg_scale = torch.tensor([4.0], device=device)
    out = dict(layers[-1], up=up, reg_scale=reg_scale)
    out["aux_outputs"] = [dict(l, teacher_corners=layers[-1]["pred_corners"], teacher_logits=layers[-1]["pred_logits"]) for l in layers[:-1]]
    out["enc_aux_outputs"] = [{"pred_logits": rnd(B, Q, C), "pred_boxes": boxes(B, Q)}]
    out["pre_outputs"] = {"pred_logits": rnd(B, Q, C), "pred_boxes": boxes(B, Q)}
    out["enc_meta"] = {"class_agnostic": False}
    out["dn_outputs"] = [dict(l, teacher_corners=dn_layers[-1]["pred_corners"], teacher_logits=dn_layers[-1]["pred_logits"]) for l in dn_layers]
    out["dn_pre_outputs"] = {"pred_logits": rnd(B, dn, C), "pred_boxes": boxes(B, dn)}
    return out


def criterion_targets_and_meta(B=2, dn=8, C=6, device="cpu"):
    """Targets with 2 GT per image and the matching dn_meta (2 groups of 2 pos + 2 neg = 8 dn queries)."""
    targets = [
        {"labels": torch.tensor([1, 4], device=device), "boxes": torch.tensor([[.3, .4, .2, .2], [.6, .5, .3, .25]], device=device)},
        {"labels": torch.tensor([2, 0], device=device), "boxes": torch.tensor([[.45, .4, .3, .2], [.7, .6, .2, .25]], device=device)},
    ][:B]
    pos = torch.tensor([0, 1, 4, 5])
    meta = {"dn_positive_idx": tuple(pos.clone() for _ in range(B)), "dn_num_group": 2, "dn_num_split": [dn, 0]}
    return targets, meta


def make_validator_case(seed, n_images=12, n_classes=5):
    """Ground truth / prediction lists in the Validator's format (absolute xyxy boxes): per image a few GT boxes;
    predictions = jittered copies (some with a wrong label), duplicates of the same object, pure false positives and
    misses; some images without GT and / or without predictions."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    gt, preds = [], []
    for i in range(n_images):
        n = int(rng.integers(0, 6)) if i % 5 else 0
        xy = rng.uniform(20, 400, (n, 2))
        wh = rng.uniform(20, 160, (n, 2))
        g_boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        g_labels = rng.integers(0, n_classes, n)
        p_boxes, p_labels, p_scores = [], [], []
        for b, l in zip(g_boxes, g_labels):
            r = rng.uniform()
            if r < 0.15:
                continue                                              # miss
            jitter = rng.normal(0, 6 if r < 0.8 else 40, 4).astype(np.float32)
            p_boxes.append(b + jitter)
            p_labels.append(l if rng.uniform() < 0.8 else (l + 1) % n_classes)
            p_scores.append(rng.uniform(0.3, 1.0))
            if rng.uniform() < 0.25:                                  # duplicate detection of the same object
                p_boxes.append(b + rng.normal(0, 3, 4).astype(np.float32))
                p_labels.append(l)
                p_scores.append(rng.uniform(0.3, 1.0))
        for _ in range(int(rng.integers(0, 3)) if i % 7 else 0):      # pure false positives
            q = rng.uniform(20, 400, 2)
            p_boxes.append(np.concatenate([q, q + rng.uniform(20, 100, 2)]).astype(np.float32))
            p_labels.append(int(rng.integers(0, n_classes)))
            p_scores.append(rng.uniform(0.3, 1.0))
        gt.append({"labels": torch.tensor(g_labels, dtype=torch.int64), "boxes": torch.tensor(g_boxes).reshape(-1, 4)})
        preds.append({"labels": torch.tensor(p_labels, dtype=torch.int64),
                      "boxes": torch.tensor(np.array(p_boxes, dtype=np.float32)).reshape(-1, 4),
                      "scores": torch.tensor(p_scores, dtype=torch.float32)})
    return gt, preds


def make_validator_mask_case(seed, n_images=10, n_classes=4, hw=(96, 128), pred_hw=None, probs=False):
    """`make_validator_case` with instance masks: ground truth = filled ellipses inside the boxes (uint8 [G, H, W]); predictions
    = the jittered boxes' ellipses, `masks` either uint8 or float probabilities (soft edge, binarised by the validator with
    > conf_thresh); `pred_hw` gives the predictions another resolution (the validator resizes them)."""
    import numpy as np
    import torch
    gt, preds = make_validator_case(seed, n_images, n_classes)
    H, W = hw
    ph, pw = pred_hw or hw

    def ellipses(boxes, h, w, soft, rng):
        ys = torch.arange(h, dtype=torch.float32)[None, :, None] + 0.5
        xs = torch.arange(w, dtype=torch.float32)[None, None, :] + 0.5
        b = boxes.clone() * torch.tensor([w / 560.0, h / 560.0, w / 560.0, h / 560.0])
        cx, cy = (b[:, 0] + b[:, 2])[:, None, None] / 2, (b[:, 1] + b[:, 3])[:, None, None] / 2
        rx, ry = ((b[:, 2] - b[:, 0]) / 2).clamp(min=1.0)[:, None, None], ((b[:, 3] - b[:, 1]) / 2).clamp(min=1.0)[:, None, None]
        d = ((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2
        if soft:
            return torch.sigmoid((1.0 - d) * 4.0 + torch.tensor(rng.normal(0, 0.3, d.shape), dtype=torch.float32))
        return (d <= 1.0).to(torch.uint8)

    rng = np.random.default_rng(seed + 100)
    for g, p in zip(gt, preds):
        g["masks"] = ellipses(g["boxes"], H, W, False, rng) if len(g["labels"]) else torch.zeros((0, H, W), dtype=torch.uint8)
        if len(p["labels"]):
            m = ellipses(p["boxes"], ph, pw, probs, rng)
        else:
            m = torch.zeros((0, ph, pw), dtype=torch.float32 if probs else torch.uint8)
        p["masks"] = m                                          # float masks are binarised by the validator with > conf_thresh
    return gt, preds
