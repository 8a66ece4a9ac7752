"""A18 post-processor (reference src/dl/export.py:61-100, src/dl/utils.py:673-712) against goldens generated from the
reference: labels / query split are integer work -> bit-exact; boxes are exact fp32 (same operation order, integral after
floor/ceil); scores within 1e-6.  CPU: the oracle restatement + the host box mapping.  GPU: the HIP kernel."""
import numpy as np
import pytest
import torch

from tests import helpers

G = helpers.GOLDEN_DIR
CASES = ((0, {}), (1, dict(B=3, Q=40, C=7)))


def _canon(labels, boxes, scores):
    """torch.topk leaves the order of tied scores unspecified: sort (score desc, label, box) lexicographically."""
    labels, boxes, scores = (np.asarray(a) for a in (labels, boxes, scores))
    out = []
    for b in range(labels.shape[0]):
        key = np.lexsort((boxes[b][:, 3], boxes[b][:, 2], boxes[b][:, 1], boxes[b][:, 0], labels[b], -scores[b]))
        out.append((labels[b][key], boxes[b][key], scores[b][key]))
    return out


def _check(got, g, prefix, k_expected):
    labels, boxes, scores = got
    assert labels.dtype == torch.int64 and boxes.dtype == torch.float32 and scores.dtype == torch.float32
    assert labels.shape[1] == k_expected
    a = _canon(labels.cpu().numpy(), boxes.cpu().numpy(), scores.cpu().numpy())
    b = _canon(g[prefix + "/labels"], g[prefix + "/boxes"], g[prefix + "/scores"])
    for (la, ba, sa), (lb, bb, sb) in zip(a, b):
        np.testing.assert_allclose(sa, sb, rtol=0, atol=1e-6)
        # scores 1 ulp apart may swap neighbours between two exp implementations: compare as multisets of rows
        ra = sorted(map(tuple, np.concatenate([la[:, None].astype(np.float64), ba], 1).tolist()))
        rb = sorted(map(tuple, np.concatenate([lb[:, None].astype(np.float64), bb], 1).tolist()))
        assert ra == rb                                      # labels and boxes bit-exact
        assert (np.diff(sa) <= 0).all()                      # descending


@pytest.mark.parametrize("seed,kw", CASES)
def test_oracle_postprocessor_matches_reference(oracle_backend, seed, kw):
    from custom_d_fine_amd.dl.export import DFINEPostProcessor
    g = np.load(f"{G}/postprocess.npz")
    logits, boxes, _ = helpers.make_postprocess_case(seed, **kw)
    pp = DFINEPostProcessor(logits.shape[-1], num_top_queries=300)
    o = {"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes)}
    for hh, ww in ((640, 640), (384, 512)):
        _check(pp(o, hh, ww), g, f"s{seed}/{hh}x{ww}", min(300, logits.shape[1] * logits.shape[2]))


@pytest.mark.parametrize("seed,kw", CASES)
@pytest.mark.parametrize("keep_ratio", [False, True])
def test_process_boxes_matches_reference(seed, kw, keep_ratio):
    from custom_d_fine_amd.dl.postprocess import process_boxes
    g = np.load(f"{G}/postprocess.npz")
    _, boxes, orig = helpers.make_postprocess_case(seed, **kw)
    got = process_boxes(torch.tensor(boxes), (640, 640), torch.tensor(orig), keep_ratio)
    np.testing.assert_allclose(got.numpy(), g[f"s{seed}/process_boxes/keep{int(keep_ratio)}"], rtol=1e-6, atol=1e-4)


def test_preds_postprocess_contract(oracle_backend):
    from custom_d_fine_amd.dl.postprocess import preds_postprocess
    logits, boxes, orig = helpers.make_postprocess_case(0)
    res = preds_postprocess(torch.zeros(2, 3, 640, 640), {"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes)},
                            torch.tensor(orig), 80, False, 0.5)
    assert len(res) == 2
    for r in res:
        assert set(r) == {"labels", "boxes", "scores", "all_boxes", "all_scores", "all_labels"}
        assert r["all_scores"].shape == (300,) and (r["scores"] >= 0.5).all() and len(r["scores"]) == int((r["all_scores"] >= 0.5).sum())
        assert r["labels"].dtype == torch.int64 and r["labels"].max() < 80


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", CASES)
def test_hip_postprocessor_matches_reference(cuda, seed, kw):
    from custom_d_fine_amd.dl.export import DFINEPostProcessor
    g = np.load(f"{G}/postprocess.npz")
    logits, boxes, _ = helpers.make_postprocess_case(seed, **kw)
    pp = DFINEPostProcessor(logits.shape[-1], num_top_queries=300)
    o = {"pred_logits": torch.tensor(logits, device=cuda), "pred_boxes": torch.tensor(boxes, device=cuda)}
    for hh, ww in ((640, 640), (384, 512)):
        _check(pp(o, hh, ww), g, f"s{seed}/{hh}x{ww}", min(300, logits.shape[1] * logits.shape[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_postprocessor_full_size_vs_oracle(cuda, dtype):
    """B = 32, Q = 300, C = 80 (the eval shape of BASELINE configs[2]); oracle = the torch restatement on the host."""
    from custom_d_fine_amd import kernels
    from oracle import torch_backend
    gen = torch.Generator().manual_seed(5)
    logits = (torch.randn(32, 300, 80, generator=gen) * 2 - 2).to(dtype)
    boxes = torch.cat([torch.rand(32, 300, 2, generator=gen), torch.rand(32, 300, 2, generator=gen) * 0.5], -1)
    labels, qidx, out_boxes, scores = kernels.detection_topk(logits.to(cuda), boxes.to(cuda), 300, 640, 640)
    rl, rq, rb, rs = torch_backend.detection_topk(logits, boxes, 300, 640, 640)
    assert (qidx < 300).all() and (labels < 80).all() and (qidx >= 0).all() and (labels >= 0).all()
    np.testing.assert_allclose(scores.cpu().numpy(), rs.numpy(), atol=1e-6, rtol=0)
    flat_got, flat_ref = (qidx * 80 + labels).cpu(), rq * 80 + rl
    for b in range(32):
        assert len(set(flat_got[b].tolist())) == 300
        if dtype == torch.float32:                            # bf16 logits tie at the cut: any of the tied is valid
            assert set(flat_got[b].tolist()) == set(flat_ref[b].tolist())
    # every returned box is the exact conversion of the returned query's box
    _, q_all, b_all, _ = torch_backend.detection_topk(torch.zeros(32, 300, 1), boxes, 300, 640, 640)   # all queries
    table = torch.empty(32, 300, 4)
    table.scatter_(1, q_all.unsqueeze(-1).expand(-1, -1, 4), b_all)
    want = table.gather(1, qidx.cpu().unsqueeze(-1).expand(-1, -1, 4))
    np.testing.assert_array_equal(out_boxes.cpu().numpy(), want.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("C,K", [(365, 300), (365, 2000), (80, 1500), (91, 5000)])
def test_hip_postprocessor_many_classes_and_large_k(cuda, C, K):
    """Objects365-sized heads (300 x 365 = 109 500 scores > the 32 768 register-resident keys), K above the 1024-entry sort
    network, and K above the kernel's range (device composition): same (query, class) set and exact boxes as the oracle."""
    from custom_d_fine_amd import kernels
    from oracle import torch_backend
    gen = torch.Generator().manual_seed(C + K)
    logits = torch.randn(3, 300, C, generator=gen) * 2 - 2
    boxes = torch.cat([torch.rand(3, 300, 2, generator=gen), torch.rand(3, 300, 2, generator=gen) * 0.5], -1)
    labels, qidx, out_boxes, scores = kernels.detection_topk(logits.to(cuda), boxes.to(cuda), K, 480, 640)
    rl, rq, rb, rs = torch_backend.detection_topk(logits, boxes, K, 480, 640)
    np.testing.assert_allclose(scores.cpu().numpy(), rs.numpy(), atol=1e-6, rtol=0)
    assert (np.diff(scores.cpu().numpy(), axis=1) <= 0).all()
    flat_got, flat_ref = (qidx * C + labels).cpu(), rq * C + rl
    for b in range(3):
        assert set(flat_got[b].tolist()) == set(flat_ref[b].tolist()) and len(set(flat_got[b].tolist())) == K
    order = torch.argsort(flat_got, 1), torch.argsort(flat_ref, 1)
    np.testing.assert_array_equal(out_boxes.cpu().gather(1, order[0].unsqueeze(-1).expand(-1, -1, 4)).numpy(),
                                  rb.gather(1, order[1].unsqueeze(-1).expand(-1, -1, 4)).numpy())


@pytest.mark.gpu
def test_hip_postprocessor_nan_logits_rank_first(cuda):
    """torch.topk ranks NaN above every number (reference export.py:73 runs it on the sigmoid scores): so does the kernel."""
    from custom_d_fine_amd import kernels
    gen = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 300, 80, generator=gen)
    logits[0, 7, 3] = float("nan")
    logits[1, 299, 79] = float("nan")
    boxes = torch.rand(2, 300, 4, generator=gen) * 0.5 + 0.25
    labels, qidx, _, scores = kernels.detection_topk(logits.to(cuda), boxes.to(cuda), 300, 640, 640)
    ref = torch.topk(torch.sigmoid(logits).flatten(1), 300, dim=-1)
    assert torch.isnan(scores[:, 0]).all() and not torch.isnan(scores[:, 1:]).any()
    assert (qidx[:, 0].cpu() * 80 + labels[:, 0].cpu()).tolist() == ref.indices[:, 0].tolist() == [7 * 80 + 3, 299 * 80 + 79]
    np.testing.assert_allclose(scores[:, 1:].cpu().numpy(), ref.values[:, 1:].numpy(), atol=1e-6, rtol=0)


# ------------------------------------------------------------------------------------------------ masks of the hand-off
def _check_process_masks(device):
    from custom_d_fine_amd.dl.postprocess import cleanup_masks, process_masks
    g = np.load(f"{G}/postprocess_masks.npz")
    pm = torch.tensor(g["pred_masks"]).to(device)
    orig = g["orig_sizes"]
    for keep in (0, 1):
        got = process_masks(pm, (80, 96), orig, bool(keep))
        for b, m in enumerate(got):
            want = g[f"process_masks/keep{keep}/{b}"].astype(np.float32)
            assert m.shape == want.shape
            assert np.abs(m.cpu().numpy() - want).max() <= 1e-3            # fixture stored as fp16
            bits = np.unpackbits(g[f"process_masks/keep{keep}/{b}_bin"], axis=-1)[..., : m.shape[-1]].astype(bool)
            flips = ((m.cpu().numpy() >= 0.5) != bits) & (np.abs(want - 0.5) > 2e-3)   # binarisation agrees away from the threshold
            assert not flips.any()
    mb = (process_masks(pm[:1], (80, 96), orig[:1], False)[0] >= 0.5).to(torch.uint8)
    got = cleanup_masks(mb, torch.tensor(g["cleanup/boxes"]).to(device)).cpu().numpy()
    want = np.unpackbits(g["cleanup/masks"], axis=-1)[..., : got.shape[-1]]
    near = np.abs(process_masks(pm[:1], (80, 96), orig[:1], False)[0].cpu().numpy() - 0.5) <= 2e-3
    assert ((got != want) & ~near).sum() == 0
    assert got[4].sum() == 0 and got[1].sum() == mb[1].sum().item()        # empty box clears everything, full-frame box nothing


def test_process_and_cleanup_masks_match_reference():
    _check_process_masks("cpu")


@pytest.mark.gpu
def test_process_and_cleanup_masks_match_reference_on_device(cuda):
    """The same goldens through the HIP bilinear kernel."""
    _check_process_masks(cuda)


def test_preds_postprocess_with_masks_contract(oracle_backend):
    from custom_d_fine_amd.dl.postprocess import gt_postprocess, preds_postprocess
    logits, boxes, orig = helpers.make_postprocess_case(0)
    B, Q = logits.shape[:2]
    torch.manual_seed(0)
    out = {"pred_logits": torch.tensor(logits), "pred_boxes": torch.tensor(boxes), "pred_masks": torch.rand(B, Q, 40, 40)}
    res = preds_postprocess(torch.zeros(B, 3, 160, 160), out, orig, logits.shape[-1], False, 0.5)
    for r, osz in zip(res, orig):
        if len(r["labels"]):
            assert r["masks"].dtype == torch.uint8 and tuple(r["masks"].shape) == (len(r["labels"]), int(osz[0]), int(osz[1]))
            x1, y1, x2, y2 = r["boxes"][0].tolist()
            m = r["masks"][0]
            assert m[: int(np.floor(y1))].sum() == 0 and m[:, : int(np.floor(x1))].sum() == 0    # nothing outside the box
    tg = [{"labels": torch.tensor([1]), "boxes": torch.tensor([[0.5, 0.5, 0.4, 0.4]]), "masks": torch.ones(1, 160, 160)},
          {"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4), "masks": torch.zeros(0, 160, 160)}]
    gt = gt_postprocess(torch.zeros(2, 3, 160, 160), tg, orig[:2], False)
    assert tuple(gt[0]["masks"].shape) == (1, int(orig[0][0]), int(orig[0][1])) and gt[0]["masks"].min() == 1
    assert tuple(gt[1]["masks"].shape) == (0, int(orig[1][0]), int(orig[1][1]))
