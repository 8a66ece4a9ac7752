"""(f3) device-side data path, first pieces (custom_d_fine_amd/dl/data_device.py, csrc/data.hip): label parsing and mosaic /
affine geometry on the host (hand-computed cases of the reference's formulas: src/dl/dataset.py:31-73, src/dl/utils.py:298-414),
the box / image kernels against the oracle restatement and size-independent properties, the multi-scale collate against
F.interpolate (what the reference calls, dataset.py:667-694)."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from custom_d_fine_amd.dl import data_device as D
from oracle import np_ref


def test_parse_yolo_label_file(tmp_path):
    p = tmp_path / "a.txt"
    p.write_text("# comment\n\n1 0.5 0.5 0.2 0.4\n0 0.1 0.1 0.3 0.1 0.3 0.5 0.1 0.5\n2 0.2 0.2 0.6 0.2 0.4 0.8 0.9\n")
    boxes, polys = D.parse_yolo_label_file(p)
    assert boxes.dtype == np.float32 and boxes.shape == (3, 5)
    np.testing.assert_allclose(boxes[0], [1, 0.5, 0.5, 0.2, 0.4])
    np.testing.assert_allclose(boxes[1], [0, 0.2, 0.3, 0.2, 0.4], atol=1e-6)          # bbox of the 4-point polygon
    assert polys[0].shape == (0, 2) and polys[1].shape == (4, 2)
    assert polys[2].shape == (3, 2)                                                 # odd trailing value dropped
    np.testing.assert_allclose(boxes[2], [2, 0.4, 0.5, 0.4, 0.6], atol=1e-6)
    (tmp_path / "e.txt").write_text("\n")
    b, pl = D.parse_yolo_label_file(tmp_path / "e.txt")
    assert b.shape == (0, 5) and pl == []
    (tmp_path / "bad.txt").write_text("0 0.1 0.2 0.3\n")
    with pytest.raises(ValueError):
        D.parse_yolo_label_file(tmp_path / "bad.txt")


def test_mosaic_coordinates_hand_cases():
    # 640 x 640 target, centre (700, 500), frames of 640 x 480 (w x h)
    assert D.get_mosaic_coordinate(0, 700, 500, 640, 480, 640, 640) == ((60, 20, 700, 500), (0, 0, 640, 480))
    assert D.get_mosaic_coordinate(1, 700, 500, 640, 480, 640, 640) == ((700, 20, 1280, 500), (0, 0, 580, 480))
    assert D.get_mosaic_coordinate(2, 700, 500, 640, 480, 640, 640) == ((60, 500, 700, 980), (0, 0, 640, 480))
    assert D.get_mosaic_coordinate(3, 700, 500, 640, 480, 640, 640) == ((700, 500, 1280, 980), (0, 0, 580, 480))
    # a centre close to the canvas corner crops the top-left frame from its bottom-right part
    assert D.get_mosaic_coordinate(0, 400, 390, 640, 480, 640, 640) == ((0, 0, 400, 390), (240, 90, 640, 480))


def test_transform_matrix_closed_form():
    M, s = D.get_transform_matrix((1280, 1280), (640, 640), 0.0, (1.0, 1.0), 0.0, 0.0, random.Random(0))
    assert s == 1.0
    np.testing.assert_allclose(M, [[1, 0, -320], [0, 1, -320], [0, 0, 1]], atol=1e-12)   # canvas centre -> target centre
    rng = random.Random(5)
    M, s = D.get_transform_matrix((1280, 1280), (640, 640), 10.0, (0.5, 1.5), 2.0, 0.1, rng)
    rng = random.Random(5)                                                         # same draws, composed by hand
    a, sc = rng.uniform(-10, 10), rng.uniform(0.5, 1.5)
    shx, shy = np.tan(np.radians(rng.uniform(-2, 2))), np.tan(np.radians(rng.uniform(-2, 2)))
    tx, ty = rng.uniform(0.4, 0.6) * 640, rng.uniform(0.4, 0.6) * 640
    assert s == sc
    ca, sa = sc * np.cos(np.radians(a)), sc * np.sin(np.radians(a))
    R = np.array([[ca, sa, 0], [-sa, ca, 0], [0, 0, 1]])
    S = np.array([[1, shx, 0], [shy, 1, 0], [0, 0, 1]])
    T = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]])
    C = np.array([[1, 0, -640], [0, 1, -640], [0, 0, 1.0]])
    np.testing.assert_allclose(M, T @ S @ R @ C, atol=1e-9)


def test_oracle_affine_boxes_hand_case():
    # pure scale 0.5 about the canvas centre (640, 640) onto a 640 x 640 target centred at (320, 320)
    M = np.array([[0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.0]])
    boxes = np.array([[100, 200, 300, 600], [0, 0, 2, 2], [1200, 1200, 1400, 1300], [10, 10, 400, 14]], dtype=np.float32)
    new, keep = np_ref.affine_boxes(boxes, M, 0.5, (640, 640))
    np.testing.assert_allclose(new, [[50, 100, 150, 300], [0, 0, 1, 1], [600, 600, 640, 640], [5, 5, 200, 7]])
    assert keep.tolist() == [True, False, True, False]      # 1-pixel box fails w > 2; the clipped one keeps 40 % of its area; 195 x 2 fails h > 2


def test_yolo_dataset_round_trip_cpu(tmp_path):
    root = D.write_synthetic_yolo_dataset(tmp_path / "ds", n_images=4, size=(120, 160), num_classes=3, seed=1)
    ds = D.YoloTxtDataset(root, img_size=(96, 128))
    assert len(ds) == 4
    images, targets = ds.batch([0, 3], "cpu")
    assert images.shape == (2, 3, 96, 128) and images.dtype == torch.float32 and 0 <= images.min() and images.max() <= 1
    for t in targets:
        assert t["labels"].dtype == torch.int64 and t["boxes"].shape[1] == 4 and len(t["labels"]) == len(t["boxes"]) >= 1
        assert t["orig_size"].tolist() == [120, 160] and (t["boxes"] > 0).all() and (t["boxes"] < 1).all()


@pytest.mark.gpu
def test_affine_boxes_kernel_matches_oracle(cuda):
    from custom_d_fine_amd import hip
    rng = random.Random(3)
    g = np.random.default_rng(3)
    for _ in range(5):
        M, s = D.get_transform_matrix((1280, 1280), (640, 640), 10.0, (0.5, 1.5), 2.0, 0.1, rng)
        xy = g.uniform(0, 1200, (200, 2)).astype(np.float32)
        wh = g.uniform(1, 400, (200, 2)).astype(np.float32)
        boxes = np.concatenate([xy, np.minimum(xy + wh, 1280)], 1)
        got, keep = hip.affine_boxes(torch.from_numpy(boxes).to(cuda), M[:2], s, (640, 640), 0.1)
        want, wkeep = np_ref.affine_boxes(boxes, M, s, (640, 640))
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-3, rtol=0)
        sure = np.abs((want[:, 2] - want[:, 0]) - 2) > 0.01                  # away from the w > 2 / h > 2 decision boundary
        sure &= np.abs((want[:, 3] - want[:, 1]) - 2) > 0.01
        assert (keep.cpu().numpy().astype(bool)[sure] == wkeep[sure]).mean() > 0.995


@pytest.mark.gpu
def test_warp_affine_kernel(cuda):
    from custom_d_fine_amd import hip
    g = np.random.default_rng(0)
    src = g.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    s_dev = torch.from_numpy(src).to(cuda)
    ident = np.array([[1, 0, 0], [0, 1, 0.0]])
    assert np.array_equal(hip.warp_affine(s_dev, ident, (90, 120)).cpu().numpy(), src)
    shift = np.array([[1, 0, 7], [0, 1, -4.0]])                                    # integer translation: a shifted copy + border
    out = hip.warp_affine(s_dev, shift, (90, 120), 114).cpu().numpy()
    assert np.array_equal(out[:86, 7:], src[4:, :113]) and (out[86:] == 114).all() and (out[:, :7] == 114).all()
    rng = random.Random(1)
    for _ in range(4):
        M, _ = D.get_transform_matrix((90, 120), (64, 80), 15.0, (0.6, 1.4), 3.0, 0.1, rng)
        got = hip.warp_affine(s_dev, M[:2], (80, 64), 114).cpu().numpy()
        want = np_ref.warp_affine_u8(src, M[:2], (80, 64), 114)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got != want).mean() < 0.01
    ramp = np.repeat(np.arange(120, dtype=np.uint8)[None, :, None], 90, 0).repeat(3, 2)   # a linear ramp stays linear under scaling
    M = np.array([[2.0, 0, 0], [0, 2.0, 0]])
    out = hip.warp_affine(torch.from_numpy(ramp).to(cuda), M, (90, 120), 0).cpu().numpy()
    np.testing.assert_allclose(out[10, 20:100:2, 0], np.arange(10, 50), atol=1)


@pytest.mark.gpu
def test_mosaic_place_kernel_matches_resize_oracle(cuda):
    from custom_d_fine_amd import hip
    g = np.random.default_rng(2)
    src = g.integers(0, 256, (60, 90, 3), dtype=np.uint8)
    canvas = torch.full((200, 240, 3), 114, dtype=torch.uint8, device=cuda)
    resized = np_ref.resize_linear_u8(src, 100, 120)                              # (rh, rw) = (100, 120)
    hip.mosaic_place(torch.from_numpy(src).to(cuda), canvas, (100, 120), (30, 20, 130, 90), (20, 30))
    out = canvas.cpu().numpy()
    assert np.array_equal(out[20:90, 30:130], resized[30:100, 20:120])
    out[20:90, 30:130] = 114
    assert (out == 114).all()


@pytest.mark.gpu
def test_multiscale_collate_matches_interpolate(cuda):
    torch.manual_seed(0)
    images = torch.rand(3, 3, 96, 128, device=cuda)
    masks = (torch.rand(4, 96, 128, device=cuda) > 0.5).to(torch.uint8)
    targets = [{"masks": masks.clone()}, {"masks": torch.zeros(0, 96, 128, dtype=torch.uint8, device=cuda)}, {}]
    for off in (-64, 32):
        out, tg = D.multiscale_collate(images, [dict(t) for t in targets], off)
        want = F.interpolate(images, size=(96 + off, 128 + off), mode="bilinear", align_corners=False)
        assert (out - want).abs().max() < 1e-5
        wm = (F.interpolate(masks.unsqueeze(1).float(), size=(96 + off, 128 + off), mode="bilinear", align_corners=False).squeeze(1) > 0.5)
        assert (tg[0]["masks"].bool() != wm).float().mean() < 1e-3 and tg[0]["masks"].dtype == torch.uint8


@pytest.mark.gpu
def test_mosaic_affine_end_to_end(cuda, tmp_path):
    root = D.write_synthetic_yolo_dataset(tmp_path / "ds", n_images=4, size=(120, 160), num_classes=3, seed=2)
    ds = D.YoloTxtDataset(root, img_size=(128, 128))
    frames, labels = [], []
    for i in range(4):
        img, boxes = ds._load(i)
        frames.append(torch.from_numpy(np.ascontiguousarray(img)).to(cuda))
        labels.append(boxes)
    image, cls, boxes = D.mosaic_affine(frames, labels, (128, 128), random.Random(4), degrees=5.0, translate=0.1, scales=(0.8, 1.2), shear=1.0)
    assert image.shape == (128, 128, 3) and image.dtype == torch.uint8 and image.is_cuda
    assert cls.dtype == torch.int64 and boxes.shape == (len(cls), 4) and len(cls) >= 1
    assert (boxes >= 0).all() and (boxes <= 128).all() and ((boxes[:, 2:] - boxes[:, :2]) > 1).all()
    # identity affine: the target frame is the centre of the mosaic canvas, and the rectangles drawn into the source frames are
    # found under their transformed boxes (the box colour of write_synthetic_yolo_dataset identifies the class)
    image, cls, boxes = D.mosaic_affine(frames, labels, (128, 128), random.Random(4), degrees=0.0, translate=0.0, scales=(1.0, 1.0), shear=0.0)
    img = image.cpu().numpy().astype(int)
    hits = 0
    for c, b in zip(cls.tolist(), boxes.cpu().numpy()):
        x0, y0, x1, y1 = [int(round(v)) for v in b]
        if x1 - x0 < 6 or y1 - y0 < 6:
            continue
        px = img[(y0 + y1) // 2, (x0 + x1) // 2]
        hits += int(np.abs(px - np.array([60 + 60 * c, 200 - 50 * c, 90 + 40 * c])).max() <= 3)
    assert hits >= max(1, int(0.6 * sum((b[2] - b[0] >= 6) and (b[3] - b[1] >= 6) for b in boxes.cpu().numpy())))


@pytest.mark.gpu
def test_dataset_batch_with_mosaic(cuda, tmp_path):
    """`batch(..., mosaic_prob=1)`: every sample is a 2 x 2 mosaic warped into the target frame - the batch contract holds
    (images in [0, 1], boxes normalised cxcywh inside the frame), the draw is reproducible, and the plain path is untouched."""
    root = D.write_synthetic_yolo_dataset(tmp_path / "ds", n_images=6, size=(120, 160), num_classes=3, seed=3)
    ds = D.YoloTxtDataset(root, img_size=(128, 160))
    im1, t1 = ds.batch([0, 1, 2], cuda, mosaic_prob=1.0, rng=random.Random(5))
    im2, t2 = ds.batch([0, 1, 2], cuda, mosaic_prob=1.0, rng=random.Random(5))
    assert im1.shape == (3, 3, 128, 160) and im1.dtype == torch.float32 and 0.0 <= im1.min() and im1.max() <= 1.0
    assert torch.equal(im1, im2) and all(torch.equal(a["boxes"], b["boxes"]) for a, b in zip(t1, t2))
    n = 0
    for t in t1:
        assert t["labels"].dtype == torch.int64 and t["boxes"].shape == (len(t["labels"]), 4) and t["orig_size"].tolist() == [128, 160]
        if len(t["labels"]):
            b = t["boxes"]
            assert (b[:, 2:] > 0).all() and (b[:, :2] - b[:, 2:] / 2 >= -1e-6).all() and (b[:, :2] + b[:, 2:] / 2 <= 1 + 1e-6).all()
            n += len(b)
    assert n >= 1
    plain, _ = ds.batch([0, 1, 2], cuda)
    plain0, _ = ds.batch([0, 1, 2], cuda, mosaic_prob=0.0, rng=random.Random(5))
    assert torch.equal(plain, plain0) and not torch.equal(plain, im1)


@pytest.mark.parametrize("n,world,bs", [(31, 2, 8), (47, 4, 4), (15, 2, 8), (1, 4, 8), (64, 8, 8), (100, 3, 7)])
def test_shard_batches_same_count_on_every_rank(n, world, bs):
    """The ranks of a data-parallel run must yield the same number of batches (every step all-reduces); sharding follows
    torch's DistributedSampler(shuffle=True, drop_last=False): one shared permutation, wrapped around to a multiple of the
    world size, strided by rank (ref dataset.py:562-568); the loader keeps the short last batch."""
    from custom_d_fine_amd.dl.train import shard_batches
    from torch.utils.data import DistributedSampler
    per_rank = [shard_batches(n, r, world, bs, seed=5) for r in range(world)]
    counts = {len(b) for b in per_rank}
    assert len(counts) == 1 and counts.pop() == -(-(-(-n // world)) // bs) >= 1
    assert all(len(b) > 0 for batches in per_rank for b in batches)
    flat = [[i for b in batches for i in b] for batches in per_rank]
    assert len({len(f) for f in flat}) == 1
    assert set(i for f in flat for i in f) == set(range(n))               # every sample is seen
    # same structure as the reference's sampler: per-rank sample count and the multiset of indices over all ranks
    ref = [list(DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, drop_last=False, seed=0)) for r in range(world)]
    assert [len(f) for f in flat] == [len(f) for f in ref]
    import collections
    assert sorted(collections.Counter(i for f in flat for i in f).values()) == sorted(collections.Counter(i for f in ref for i in f).values())
    assert shard_batches(0, 0, world, bs, 1) == []


def test_trainer_epoch_schedule_matches_batches(tmp_path):
    """steps_per_epoch (the OneCycleLR length) is the number of batches a rank really runs, also when the folder does not
    divide by world * batch size."""
    from PIL import Image
    from custom_d_fine_amd.dl import train as T
    (tmp_path / "images").mkdir(); (tmp_path / "labels").mkdir()
    rs = np.random.RandomState(0)
    for i in range(7):
        Image.fromarray(rs.randint(0, 255, (40, 48, 3), dtype=np.uint8)).save(tmp_path / "images" / f"{i}.png")
        (tmp_path / "labels" / f"{i}.txt").write_text("0 0.5 0.5 0.2 0.2\n")
    from custom_d_fine_amd.dl.data_device import YoloTxtDataset
    n = len(YoloTxtDataset(str(tmp_path), [64, 64]))
    assert n == 7
    for world in (1, 2, 4):
        for rank in range(world):
            assert len(T.shard_batches(n, rank, world, 2, 3)) == -(-(-(-n // world)) // 2)


# ---- goldens from the reference's own functions (tests/golden/data_device.npz, tools/gen_golden.py::gen_data_device):
# `random_affine` with its matrix draw pinned, `box_candidates`, `get_mosaic_coordinate` (src/dl/utils.py:283-414)
def _golden():
    from tests import helpers
    return np.load(f"{helpers.GOLDEN_DIR}/data_device.npz")


def test_oracle_box_candidates_and_mosaic_coordinates_match_reference():
    g = _golden()
    assert np.array_equal(np_ref.box_candidates(g["cand/box1"], g["cand/box2"], area_thr=0.1), g["cand/keep"])
    for idx, xc, yc, w, h, *want in g["mosaic/rows"].tolist():
        big, small = D.get_mosaic_coordinate(idx, xc, yc, w, h, 640, 640)
        assert list(big) + list(small) == want


def test_oracle_affine_boxes_match_reference_random_affine():
    g = _golden()
    for c in range(int(g["n_affine"])):
        tin, tout = g[f"affine{c}/targets_in"], g[f"affine{c}/targets_out"]
        new, keep = np_ref.affine_boxes(tin[:, 1:5], g[f"affine{c}/M"], float(g[f"affine{c}/scale"]), tuple(g[f"affine{c}/target_size"]))
        assert int(keep.sum()) == len(tout) and 0 < len(tout) < len(tin)
        np.testing.assert_allclose(new[keep], tout[:, 1:5], rtol=0, atol=1e-4)
        assert np.array_equal(tin[keep, 0], tout[:, 0])


@pytest.mark.gpu
def test_affine_boxes_kernel_matches_reference_random_affine(cuda):
    from custom_d_fine_amd import hip
    g = _golden()
    for c in range(int(g["n_affine"])):
        tin, tout = g[f"affine{c}/targets_in"], g[f"affine{c}/targets_out"]
        M, s, size = g[f"affine{c}/M"], float(g[f"affine{c}/scale"]), tuple(int(v) for v in g[f"affine{c}/target_size"])
        got, keep = hip.affine_boxes(torch.from_numpy(tin[:, 1:5].copy()).to(cuda), M[:2], s, size, 0.1)
        keep = keep.cpu().numpy().astype(bool)
        want_new, want_keep = np_ref.affine_boxes(tin[:, 1:5], M, s, size)
        # boxes on the filter's decision boundaries aside (fp32 kernel against the reference's float64 matrix product), the
        # kept set and the kept boxes are the reference's
        w, h = want_new[:, 2] - want_new[:, 0], want_new[:, 3] - want_new[:, 1]
        sure = (np.abs(w - 2) > 0.01) & (np.abs(h - 2) > 0.01)
        assert np.array_equal(keep[sure], want_keep[sure])
        both = keep & want_keep
        np.testing.assert_allclose(got.cpu().numpy()[both], want_new[both], rtol=0, atol=2e-3)
        assert abs(int(keep.sum()) - len(tout)) <= int((~sure).sum())
