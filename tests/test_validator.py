"""(f2) evaluation metrics: `custom_d_fine_amd.dl.validator.Validator` against goldens generated from the reference's
`Validator` (tools/gen_golden.py::gen_validator, box path, compute_maps=False) and against the known-answer cases of the
reference's own self-test (`src/dl/validator.py:727-800`, restated with boxes instead of masks)."""
import numpy as np
import pytest
import torch

from custom_d_fine_amd.dl.validator import Validator, coco_map
from tests import helpers

G = helpers.GOLDEN_DIR


def _check_box_metrics(seed, thr, device):
    g = np.load(f"{G}/validator.npz")
    k = f"s{seed}_t{int(thr * 100)}"
    gt, preds = helpers.make_validator_case(seed)
    gt = [{n: t.to(device) for n, t in d.items()} for d in gt]
    preds = [{n: t.to(device) for n, t in d.items()} for d in preds]
    v = Validator(gt, preds, {i: f"c{i}" for i in range(5)}, conf_thresh=0.5, iou_thresh=thr, compute_maps=False)
    m = v.compute_metrics(extended=True)
    for name in ("TPs", "FPs", "FNs"):
        assert m[name] == int(g[f"{k}/{name}"]), name                      # integer work: exact
    for name in ("f1", "precision", "recall"):
        assert abs(m[name] - float(g[f"{k}/{name}"])) < 1e-12, name
    assert abs(m["iou"] - float(g[f"{k}/iou"])) < 1e-6
    assert np.array_equal(v.conf_matrix, g[f"{k}/conf_matrix"])
    assert sorted(v.class_to_idx) == g[f"{k}/classes"].tolist()
    ext = m["extended_metrics"]
    assert sorted(ext) == g[f"{k}/ext_keys"].tolist()
    np.testing.assert_allclose([float(ext[x]) for x in sorted(ext)], g[f"{k}/ext_vals"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_box_metrics_match_reference(seed, thr):
    _check_box_metrics(seed, thr, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_box_metrics_match_reference_on_device(cuda, seed, thr):
    """The same reference goldens with device-resident predictions / ground truth (the batched IoU pass runs on the GPU)."""
    _check_box_metrics(seed, thr, cuda)


def _sample(boxes, labels, scores=None):
    out = {"boxes": torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), "labels": torch.tensor(labels, dtype=torch.int64)}
    if scores is not None:
        out["scores"] = torch.tensor(scores, dtype=torch.float32)
    return out


def test_known_answer_cases_of_the_reference_self_test():
    names = {0: "class_0", 1: "class_1"}
    box = [1.0, 1.0, 3.0, 3.0]
    # perfect match
    m = Validator([_sample([box], [0])], [_sample([box], [0], [1.0])], names, compute_maps=False).compute_metrics()
    assert m["precision"] == 1.0 and m["recall"] == 1.0 and abs(m["iou"] - 1.0) < 1e-6
    # partial match above the threshold: 4x4 vs 4x3 -> IoU 0.75
    m = Validator([_sample([[0, 0, 4, 4]], [0])], [_sample([[0, 0, 4, 3]], [0], [1.0])], names, compute_maps=False).compute_metrics()
    assert m["precision"] == 1.0 and m["recall"] == 1.0 and abs(m["iou"] - 0.75) < 1e-6
    # misclassification: FP for the predicted class, FN for the GT class
    m = Validator([_sample([box], [0])], [_sample([box], [1], [1.0])], names, compute_maps=False).compute_metrics()
    assert m["precision"] == 0.0 and m["recall"] == 0.0 and m["iou"] == 0.0 and m["FPs"] == 1 and m["FNs"] == 1
    # pure false positive
    m = Validator([_sample([], [])], [_sample([box], [0], [1.0])], names, compute_maps=False).compute_metrics()
    assert m["precision"] == 0.0 and m["recall"] == 0.0 and m["FPs"] == 1 and m["FNs"] == 0
    # nothing at all
    m = Validator([_sample([], [])], [_sample([], [], [])], names, compute_maps=False).compute_metrics()
    assert m["TPs"] == m["FPs"] == m["FNs"] == 0 and m["f1"] == 0


def test_coco_map_properties():
    """mAP restatement (unpinned against torchmetrics, absent here): 1.0 for perfect detections, invariant to extra low-score
    false positives ranked after every true positive only in recall, lower with a misplaced box, -1 without ground truth."""
    gt, _ = helpers.make_validator_case(4)
    perfect = [{"boxes": g["boxes"].clone(), "labels": g["labels"].clone(), "scores": torch.full((len(g["labels"]),), 0.9)} for g in gt]
    m = coco_map(gt, perfect)
    assert abs(m["map"] - 1.0) < 1e-9 and abs(m["map_50"] - 1.0) < 1e-9
    shifted = [{"boxes": p["boxes"] + 8.0, "labels": p["labels"], "scores": p["scores"]} for p in perfect]
    m2 = coco_map(gt, shifted)
    assert m2["map"] < m["map"] and m2["map_50"] <= 1.0
    noisy = [{"boxes": torch.cat([p["boxes"], torch.tensor([[500.0, 500.0, 520.0, 520.0]])]),
              "labels": torch.cat([p["labels"], torch.tensor([0])]), "scores": torch.cat([p["scores"], torch.tensor([0.05])])} for p in perfect]
    assert abs(coco_map(gt, noisy)["map_50"] - 1.0) < 1e-9
    assert coco_map([_sample([], [])], [_sample([[1, 1, 2, 2]], [0], [0.5])])["map"] == -1.0
    full = Validator(gt, perfect, {i: f"c{i}" for i in range(5)}).compute_metrics()
    assert abs(full["mAP_50_95"] - 1.0) < 1e-9 and full["f1"] == 1.0


def test_coco_map_hand_computed_case():
    """COCO AP worked by hand (101-point interpolated precision, IoU 0.50:0.05:0.95).  One image, one class, two ground
    truths A = [0,0,10,10], B = [20,20,30,30]; detections by score: d1 = A exactly (0.9), d2 = [100,100,110,110] (0.8, a
    false positive), d3 = [20,20,30,28] (0.7: IoU with B = 80/100 = 0.8).
      IoU thresholds 0.50 .. 0.80 (7 of them): TP, FP, TP -> precision at recall 0.5 is 1, at recall 1.0 is 2/3;
        101-point AP = (51 * 1 + 50 * 2/3) / 101
      IoU thresholds 0.85, 0.90, 0.95: TP, FP, FP -> recall stops at 0.5: AP = 51 / 101."""
    gt = [_sample([[0, 0, 10, 10], [20, 20, 30, 30]], [0, 0])]
    pr = [_sample([[0, 0, 10, 10], [100, 100, 110, 110], [20, 20, 30, 28]], [0, 0, 0], [0.9, 0.8, 0.7])]
    m = coco_map(gt, pr)
    ap_lo, ap_hi = (51 + 50 * 2 / 3) / 101, 51 / 101
    assert abs(m["map_50"] - ap_lo) < 1e-9
    assert abs(m["map"] - (7 * ap_lo + 3 * ap_hi) / 10) < 1e-9


@pytest.mark.gpu
def test_device_resident_boxes(cuda):
    gt, preds = helpers.make_validator_case(2)
    gt_d = [{k: v.to(cuda) for k, v in g.items()} for g in gt]
    pr_d = [{k: v.to(cuda) for k, v in p.items()} for p in preds]
    a = Validator(gt, preds, {i: f"c{i}" for i in range(5)}).compute_metrics()
    b = Validator(gt_d, pr_d, {i: f"c{i}" for i in range(5)}).compute_metrics()
    for k in a:
        assert abs(float(a[k]) - float(b[k])) < 1e-6, k


# ------------------------------------------------------------------------------------------------ instance masks
MASK_CASES = {"dense": {}, "probs": {"probs": True}, "resized": {"pred_hw": (48, 64)}, "resized_probs": {"pred_hw": (60, 80), "probs": True}}


def _check_mask_metrics(name, seed, device):
    g = np.load(f"{G}/validator_masks.npz")
    k = f"{name}_s{seed}"
    gt, preds = helpers.make_validator_mask_case(seed, **MASK_CASES[name])
    gt = [{n: t.to(device) for n, t in d.items()} for d in gt]
    preds = [{n: t.to(device) for n, t in d.items()} for d in preds]
    v = Validator(gt, preds, {i: f"c{i}" for i in range(5)}, conf_thresh=0.5, iou_thresh=0.5, compute_maps=False)
    assert v.use_masks
    m = v.compute_metrics(extended=True)
    for nm in ("TPs", "FPs", "FNs"):
        assert m[nm] == int(g[f"{k}/{nm}"]), nm
    for nm in ("f1", "precision", "recall"):
        assert abs(m[nm] - float(g[f"{k}/{nm}"])) < 1e-12, nm
    assert abs(m["iou"] - float(g[f"{k}/iou"])) < 1e-7
    assert np.array_equal(v.conf_matrix, g[f"{k}/conf_matrix"])
    ext = m["extended_metrics"]
    assert sorted(ext) == g[f"{k}/ext_keys"].tolist()
    np.testing.assert_allclose([float(ext[x]) for x in sorted(ext)], g[f"{k}/ext_vals"], rtol=0, atol=1e-7)
    if name == "dense":                                    # the IoU matrices themselves: integer counts + one fp32 division
        for i, iou in enumerate(v.mask_ious()):
            assert np.array_equal(np.asarray(iou, dtype=np.float32), g[f"{k}/iou{i}"]), i
    # box-only metrics of the same lists are still available (ignore_masks, validator.py:109-113)
    mb = v.compute_metrics(ignore_masks=True)
    assert mb["TPs"] + mb["FNs"] == m["TPs"] + m["FNs"]


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("name", list(MASK_CASES))
def test_mask_metrics_match_reference(name, seed):
    _check_mask_metrics(name, seed, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("name", list(MASK_CASES))
def test_mask_metrics_match_reference_on_device(cuda, name, seed):
    """Device-resident masks: bit-packed (dfine_mask_pack_bits) and intersected with popcounts (dfine_mask_iou_bits), resized
    with the HIP bilinear kernel - against the reference's goldens; the IoU matrices of the dense case bit for bit."""
    _check_mask_metrics(name, seed, cuda)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(64, 64), (37, 53), (640, 640), (5, 3)])
def test_packed_mask_iou_matches_matmul_route(cuda, dtype, hw):
    """Ragged sizes (H * W not a multiple of 4 / 256), empty masks, full masks, all storage types."""
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.dl.validator import pairwise_mask_iou
    torch.manual_seed(hw[0] * 7 + hw[1])
    pm = torch.rand(7, *hw, device=cuda)
    gm = torch.rand(5, *hw, device=cuda)
    pm[0] = 0
    pm[1] = 1
    gm[0] = 0
    gm[1] = 1
    if dtype == torch.uint8:
        a, b = (pm > 0.6).to(torch.uint8), (gm > 0.4).to(torch.uint8)
    else:
        a, b = pm.to(dtype), gm.to(dtype)
    got = pairwise_mask_iou(a, b, 0.5)
    want = pairwise_mask_iou(a.cpu().float() if dtype != torch.uint8 else a.cpu(), b.cpu().float() if dtype != torch.uint8 else b.cpu(), 0.5)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert got[0].max() == 0 and got[1, 1] == 1.0
    assert hip.mask_pack_bits(a).shape == (7, 4 * ((hw[0] * hw[1] + 255) // 256))


def test_mask_map_is_reported_and_bounded():
    gt, preds = helpers.make_validator_mask_case(3)
    m = Validator(gt, preds, {i: f"c{i}" for i in range(5)}, compute_maps=True).compute_metrics()
    assert 0.0 <= m["mAP_50_95_mask"] <= m["mAP_50_mask"] <= 1.0
    perfect = [{"labels": g["labels"], "boxes": g["boxes"], "masks": g["masks"], "scores": torch.ones(len(g["labels"]))} for g in gt]
    mp = Validator(gt, perfect, {i: f"c{i}" for i in range(5)}, compute_maps=True).compute_metrics()
    assert abs(mp["mAP_50_95_mask"] - 1.0) < 1e-9 and mp["FPs"] == 0 and mp["FNs"] == 0
