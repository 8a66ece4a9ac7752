"""A3 parity: fused max-over-classes + top-K kernel vs torch (index work: bit-exact when scores are
distinct; ties resolved by ascending index)."""
import pytest
import torch

from custom_d_fine_amd import kernels

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,Q,C,K,dtype", [(4, 8400, 80, 300, torch.float32), (3, 8400, 80, 300, torch.bfloat16),
                                            (2, 500, 3, 300, torch.float32), (1, 16384, 5, 1024, torch.float32),
                                            (2, 777, 1, 10, torch.float32)])
def test_topk_matches_torch(cuda, B, Q, C, K, dtype):
    torch.manual_seed(Q + K)
    logits = (torch.randn(B, Q, C, device=cuda) * 3).to(dtype)
    idx = kernels.topk_anchors(logits, K)
    score = logits.float().max(-1).values
    ref_v, ref_i = torch.topk(score, K, dim=-1)
    got_v = score.gather(1, idx)
    assert idx.dtype == torch.int64 and idx.shape == (B, K)
    assert torch.equal(got_v, ref_v)                       # same multiset of values, same (descending) order
    if dtype == torch.float32:                              # distinct scores -> identical indices
        assert torch.equal(idx, ref_i)
    else:                                                   # bf16 scores collide: sets agree up to ties at the cut
        for b in range(B):
            cut = ref_v[b, -1]
            assert set(idx[b][got_v[b] > cut].tolist()) == set(ref_i[b][ref_v[b] > cut].tolist())
            assert (idx[b].sort().values.diff() > 0).all()


def test_topk_ties_and_strided_view(cuda):
    logits = torch.zeros(2, 600, 4, device=cuda)
    logits[:, 50:55, 1] = 1.0                               # 5 clear winners, then a 595-way tie
    idx = kernels.topk_anchors(logits, 300)
    assert idx[:, :5].tolist() == [[50, 51, 52, 53, 54]] * 2
    assert idx[0, 5:].tolist() == [i for i in range(600) if not 50 <= i < 55][:295]      # ascending index on ties
    wide = torch.randn(2, 700, 8, device=cuda)
    view = wide[:, 100:, :]                                 # non-contiguous batch stride
    ref = torch.topk(view.max(-1).values, 64, dim=-1).indices
    assert torch.equal(kernels.topk_anchors(view, 64), ref)
