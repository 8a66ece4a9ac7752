"""A11/A12 parity: device cost blocks vs the oracle (fp32 tolerance) and device LSAP vs SciPy's
results (golden) / the C oracle - indices must be bit-identical on the same cost matrix."""
import numpy as np
import pytest
import torch

from custom_d_fine_amd import hip as hipmod
from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.configs import models
from custom_d_fine_amd.d_fine.matcher import HungarianMatcher
from oracle import np_ref
from tests import helpers

pytestmark = pytest.mark.gpu
G = helpers.GOLDEN_DIR


def _device_lsap(cost_qt, cuda):
    """cost [Q, T] -> scipy-style (rows, cols) through dfine_lsap."""
    q, t = cost_qt.shape
    c = torch.tensor(np.ascontiguousarray(cost_qt.T), device=cuda)[None, None]       # [1,1,T,Q]
    cols = hipmod.lsap(c, [t]).cpu().numpy()[0]
    tt = np.nonzero(cols >= 0)[0]
    order = np.argsort(cols[tt], kind="stable")
    return cols[tt][order].astype(np.int64), tt[order].astype(np.int64)


def test_lsap_bit_exact_on_scipy_golden(cuda):
    g = np.load(f"{G}/lsap.npz")
    names = sorted({k.split("/")[0] for k in g.files if "/" in k})
    for n in names:
        r, c = _device_lsap(g[n + "/cost"], cuda)
        assert np.array_equal(r, g[n + "/rows"]) and np.array_equal(c, g[n + "/cols"]), n


def test_lsap_differential_vs_c_oracle(cuda):
    rng = np.random.default_rng(11)
    for trial in range(120):
        q = int(rng.integers(1, 330))
        t = int(rng.integers(1, 120))
        kind = trial % 4
        if kind == 0:
            c = rng.random((q, t)).astype(np.float32)
        elif kind == 1:
            c = rng.integers(0, 3, (q, t)).astype(np.float32)
        elif kind == 2:
            c = rng.integers(0, 2, (q, t)).astype(np.float32)
        else:
            c = np.round(rng.random((q, t)), 1).astype(np.float32)
        r, k = _device_lsap(c, cuda)
        ro, ko = np_ref.lsap(c)
        assert np.array_equal(r, ro) and np.array_equal(k, ko), (trial, q, t)


def test_batched_problems_one_launch(cuda):
    """K heads x B images of different target counts (incl. 0 and T > Q) in one launch."""
    rng = np.random.default_rng(5)
    K, B, Q = 3, 5, 40
    sizes = [7, 0, 55, 1, 40]
    tmax = max(sizes)
    cost = rng.random((K, B, tmax, Q)).astype(np.float32)
    cols = hipmod.lsap(torch.tensor(cost, device=cuda), sizes).cpu().numpy()
    offs = np.cumsum([0] + sizes)
    for k in range(K):
        for b, n in enumerate(sizes):
            if n == 0:
                continue
            r, c = np_ref.lsap(cost[k, b, :n, :].T)            # [Q, n]
            got = cols[k, offs[b]:offs[b + 1]]
            want = np.full(n, -1)
            want[c] = r
            assert np.array_equal(got, want), (k, b)


@pytest.mark.parametrize("seed,kw", [(0, {}), (1, dict(B=2, Q=300, C=80, sizes=(7, 23))), (2, dict(B=2, Q=6, C=4, sizes=(9, 2)))])
def test_matcher_matches_reference_golden(cuda, seed, kw):
    g = np.load(f"{G}/matcher.npz")
    logits, boxes, targets = helpers.make_matcher_case(seed, **kw)
    tg = [{k: v.to(cuda) for k, v in t.items()} for t in targets]
    sizes = [len(t["labels"]) for t in targets]
    cols, cost = kernels.hungarian_assign(
        torch.tensor(logits, device=cuda)[None], torch.tensor(boxes, device=cuda)[None],
        torch.cat([t["labels"] for t in tg]), torch.cat([t["boxes"] for t in tg]), sizes,
        2.0, 5.0, 2.0, 0.25, 2.0)
    cost = cost.cpu().numpy()[0]
    for b, n in enumerate(sizes):
        if n:
            np.testing.assert_allclose(cost[b, :, :n], g[f"s{seed}/cost{b}"], rtol=5e-5, atol=2e-4)
    matcher = HungarianMatcher(**models["m"]["matcher"])
    res = matcher({"pred_logits": torch.tensor(logits, device=cuda), "pred_boxes": torch.tensor(boxes, device=cuda)}, tg)["indices"]
    for b, (i, j) in enumerate(res):
        assert not i.is_cuda and i.dtype == torch.int64
        assert np.array_equal(i.numpy(), g[f"s{seed}/rows{b}"]) and np.array_equal(j.numpy(), g[f"s{seed}/cols{b}"])


def test_nan_costs_become_one(cuda):
    logits, boxes, targets = helpers.make_matcher_case(3, B=1, Q=20, C=4, sizes=(5,))
    logits[0, 3, :] = np.nan
    boxes[0, 7, :] = np.nan
    tg = [{k: v.to(cuda) for k, v in t.items()} for t in targets]
    cols, cost = kernels.hungarian_assign(torch.tensor(logits, device=cuda)[None], torch.tensor(boxes, device=cuda)[None],
                                          tg[0]["labels"], tg[0]["boxes"], [5], 2.0, 5.0, 2.0, 0.25, 2.0)
    cost = cost.cpu().numpy()[0, 0]
    assert (cost[3] == 1.0).all() and (cost[7] == 1.0).all() and np.isfinite(cost).all()
    r, c = np_ref.lsap(cost)
    want = np.full(5, -1)
    want[c] = r
    assert np.array_equal(cols.cpu().numpy()[0], want)


def test_full_size_six_heads(cuda):
    """BASELINE configs[2] shapes: 6 heads x 32 images x 300 queries, COCO-like target counts.
    Device indices == C oracle on the device's own cost blocks; every target gets its own query."""
    rng = np.random.default_rng(0)
    K, B, Q, C = 6, 32, 300, 80
    sizes = [int(min(max(rng.poisson(7.3), 1), 100)) for _ in range(B)]
    sizes[3] = 100
    targets = helpers.make_targets(B, C, seed=1)
    targets = []
    for n in sizes:
        targets.append({"labels": torch.from_numpy(rng.integers(0, C, n)).long().to(cuda),
                        "boxes": torch.from_numpy(np.concatenate([rng.uniform(.2, .8, (n, 2)), rng.uniform(.05, .35, (n, 2))], 1).astype(np.float32)).to(cuda)})
    logits = torch.tensor(rng.normal(-2, 2, (K, B, Q, C)).astype(np.float32), device=cuda)
    boxes = torch.tensor(np.concatenate([rng.uniform(.1, .9, (K, B, Q, 2)), rng.uniform(.02, .5, (K, B, Q, 2))], -1).astype(np.float32), device=cuda)
    cols, cost = kernels.hungarian_assign(logits, boxes, torch.cat([t["labels"] for t in targets]),
                                          torch.cat([t["boxes"] for t in targets]), sizes, 2.0, 5.0, 2.0, 0.25, 2.0)
    cols, cost = cols.cpu().numpy(), cost.cpu().numpy()
    offs = np.cumsum([0] + sizes)
    for k in range(K):
        for b, n in enumerate(sizes):
            got = cols[k, offs[b]:offs[b + 1]]
            assert len(set(got.tolist())) == n and got.min() >= 0 and got.max() < Q
            r, c = np_ref.lsap(cost[k, b, :, :n])
            want = np.empty(n, np.int64)
            want[c] = r
            assert np.array_equal(got, want), (k, b)
