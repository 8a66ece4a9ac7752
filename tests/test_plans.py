"""A13 on the device (csrc/plans.hip): the criterion's index bookkeeping without the host round trip.
* oracle pin (CPU): `oracle.np_ref.aten_argsort_desc` - the restated libstdc++ introsort behind ATen's CPU
  `torch.argsort(counts, descending=True)` (the tie order of the reference's GO vote, src/d_fine/dfine_criterion.py:570-591) -
  against torch itself, including the heapsort fallback;
* device plans (GPU): per-head gather plans, the GO union (exact sequence) and its length against the host path
  (`_cols_to_matchings` + `_get_go_indices`: the reference's algorithm on the copied-back matching), integer-exact;
* device scalar table (GPU): bit-equal to the Python arithmetic of `_forward_fused`;
* the whole criterion (GPU): losses and gradients of the synchronisation-free path against the host-plan path."""
import numpy as np
import pytest
import torch

from oracle.np_ref import aten_argsort_desc


def _musser_killer(n):
    k = n // 2
    a = [0] * n
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return np.asarray(a[:n], dtype=np.int64)


def test_oracle_argsort_matches_torch_cpu():
    rs = np.random.RandomState(3)
    cases = []
    for trial in range(1500):
        n = rs.randint(0, 700) if trial % 3 else rs.randint(0, 40)
        hi = rs.choice([2, 3, 7, 50, 100000])
        cases.append(rs.randint(1, hi + 1, size=n).astype(np.int64))
    for n in (17, 33, 100, 512, 2000):
        cases += [np.arange(n), np.arange(n)[::-1].copy(), np.ones(n, dtype=np.int64), np.arange(n) % 2,
                  _musser_killer(n), 3 * n - _musser_killer(n)]           # the last two reach the heapsort fallback
    for c in cases:
        c = np.asarray(c, dtype=np.int64)
        want = torch.argsort(torch.from_numpy(c), descending=True).numpy()
        assert np.array_equal(aten_argsort_desc(c), want), c[:40]


def _random_matchings(rs, K, sizes, Q, agree):
    """cols [K, T]: per head and image a random injection targets -> queries; with probability `agree` a head repeats
    head 0's choice for a target (repeated pairs = multiplicities > 1, ties between targets of one query)."""
    T = int(sum(sizes))
    cols = np.zeros((K, T), dtype=np.int32)
    off = 0
    for n in sizes:
        base = rs.choice(Q, size=n, replace=False)
        for k in range(K):
            if k == 0:
                q = base.copy()
            else:
                q = base.copy()
                redo = rs.rand(n) > agree
                pool = rs.permutation(Q)
                used = set(q[~redo].tolist())
                fresh = [p for p in pool if p not in used][: int(redo.sum())]
                q[redo] = fresh
                if rs.rand() < 0.5 and n > 1:                    # swap two targets' queries: same queries, other targets
                    i, j = rs.choice(n, 2, replace=False)
                    q[i], q[j] = q[j], q[i]
            cols[k, off: off + n] = q
        off += n
    return cols


@pytest.mark.gpu
@pytest.mark.parametrize("K,B,Q,tmax,agree", [(6, 32, 300, 16, 0.7), (6, 8, 300, 100, 0.5), (8, 4, 300, 300, 0.6), (3, 5, 50, 9, 0.3),
                                              (6, 16, 300, 40, 0.95), (1, 3, 20, 5, 1.0)])
def test_device_plans_equal_host_plans(cuda, K, B, Q, tmax, agree):
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.d_fine.dfine_criterion import _Plan
    from custom_d_fine_amd.d_fine.matcher import _cols_to_matchings
    crit = dfine.build_loss("n", 80, 0.0, False)
    rs = np.random.RandomState(K * 1000 + B + tmax)
    for trial in range(6):
        sizes = [int(rs.randint(0, tmax + 1)) for _ in range(B)]
        sizes[rs.randint(B)] = tmax
        if trial == 0:
            sizes[0] = 0
        assert hip.criterion_plans_supported(K, tmax, Q)
        cols = _random_matchings(rs, K, sizes, Q, agree)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        T = int(offs[-1])
        host = _cols_to_matchings(cols, sizes)
        go_host = crit._get_go_indices(host[0], host[1:])
        want_go = _Plan.pack(go_host, offs.tolist())
        hp, gp, gc, gf = hip.criterion_plans(torch.from_numpy(cols).to(cuda), torch.from_numpy(offs).to(cuda), sizes, Q,
                                             want_float_count=True)
        n = int(gc.item())
        assert n == want_go.shape[1] and float(gf.item()) == float(n)
        assert np.array_equal(gp[:, :n].cpu().numpy(), want_go), "GO union differs from the reference's order"
        for k in range(K):
            want = _Plan.pack(host[k], offs.tolist())                    # (image, query)-ordered
            got = hp[k].cpu().numpy()
            assert got.shape == (3, T)
            order = np.lexsort((got[1], got[0]))
            assert np.array_equal(got[:, order], want)


@pytest.mark.gpu
def test_device_scales_equal_python_arithmetic(cuda):
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.d_fine.arch.utils import upload
    rs = np.random.RandomState(0)
    for trial in range(20):
        world = int(rs.choice([1, 1, 2, 8]))
        go_local = int(rs.randint(0, 500))
        go_sum = go_local if world == 1 else go_local + int(rs.randint(0, 500 * (world - 1)))
        b = int(rs.choice([2, 8, 32]))
        rows, want = [], []
        num_boxes_go = max(float(np.float32(go_sum) / np.float32(world)), 1.0)
        num_pos = num_neg = 0.0
        for r in range(11):
            uses_go, teacher, is_dn = r < 6, bool(rs.rand() < 0.6), r >= 6
            q = 300 if not is_dn else 192
            n_cls = float(rs.randint(1, 400))
            n_box = float(rs.randint(1, 400))
            w = [float(x) for x in (1.0, 5.0, 2.0, 0.15, 1.5)]          # vfl, bbox, giou, fgl, ddf
            cnt_dn = int(rs.randint(0, b * 96))
            rows.append((w[0] / n_cls, 1.0 if uses_go else 0.0, n_box, w[1], w[2], w[3], w[4], 1.0 if teacher else 0.0,
                         1.0 if is_dn else 0.0, 4.0 * b * q, 4.0 * cnt_dn, 8.0 / b))
            nb = num_boxes_go if uses_go else n_box
            c_pos = c_neg = 0.0
            if teacher:                                                  # the arithmetic of DFINECriterion._forward_fused
                rows_pos = 4.0 * (go_local if uses_go else cnt_dn)
                rows_neg = 4.0 * (b * q) - rows_pos
                if not is_dn:
                    scale = 8.0 / b
                    num_pos, num_neg = (rows_pos * scale) ** 0.5, (rows_neg * scale) ** 0.5
                den = num_pos + num_neg
                c_pos = w[4] * num_pos / (den * rows_pos) if rows_pos > 0 else 0.0
                c_neg = w[4] * num_neg / (den * rows_neg) if rows_neg > 0 else 0.0
            want.append([np.float32(v) for v in (w[0] / n_cls, w[1] / nb, w[2] / nb, w[3] / nb, c_pos, c_neg)])
        params = upload(np.asarray(rows, dtype=np.float64), cuda)
        gc = torch.tensor([go_local], device=cuda, dtype=torch.int32)
        gs = None if world == 1 else torch.tensor([float(go_sum)], device=cuda)
        got = hip.criterion_scales(params, gc, gs, world).cpu().numpy()
        want = np.asarray(want, dtype=np.float32)
        if not np.isfinite(want).all():                                  # den == 0 corner (no teacher rows at all): same NaN pattern
            assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[np.isfinite(want)], want[np.isfinite(want)]), (trial, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mask", [("n", False), ("m", False), ("n", True)])
def test_criterion_without_host_sync_equals_host_plan_path(cuda, name, mask):
    """Losses and gradients of the synchronisation-free criterion against the path that copies the matching to the host."""
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.d_fine import dfine_criterion as DC
    from custom_d_fine_amd.d_fine.arch import utils as U
    from custom_d_fine_amd.dl.synthetic import make_batch
    torch.manual_seed(0)
    model = dfine.build_model(name, 80, mask, str(cuda), img_size=[320, 320]).train()
    crit = dfine.build_loss(name, 80, 0.0, mask)
    images, targets = make_batch(4, 320, seed=5, device=cuda, with_masks=mask)
    res = {}
    for dev_plans in (False, True):
        DC._DEVICE_PLANS[0] = dev_plans
        try:
            U.set_denoising_generator(torch.Generator().manual_seed(3))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(images, targets=targets)
            U.set_denoising_generator(None)
            heads = [out["pred_logits"], out["pred_boxes"], out["pred_corners"]] + [a["pred_logits"] for a in out["aux_outputs"]]
            for h in heads:
                h.retain_grad()
            losses = crit(out, targets)
            total = sum(losses.values())
            total.backward()
            res[dev_plans] = ({k: v.item() for k, v in losses.items()}, [h.grad.float().clone() for h in heads])
            model.zero_grad(set_to_none=True)
        finally:
            DC._DEVICE_PLANS[0] = True
    le, ld = res[False][0], res[True][0]
    assert set(le) == set(ld)
    for k in le:
        assert abs(le[k] - ld[k]) <= 2e-5 * max(1.0, abs(le[k])), (k, le[k], ld[k])
    for a, b in zip(res[False][1], res[True][1]):
        assert (a - b).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1e-6)
