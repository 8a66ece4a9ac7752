"""A4 parity: the one-launch denoising-group kernel (csrc/cdn.hip) against the torch composition it restates
(custom_d_fine_amd/d_fine/arch/utils.py::_cdn_group_torch, itself pinned to the reference through the n320 / s320 train-step
goldens with injected noise): class ids and logit-space boxes BIT-identical for the same random draws, with the device RNG and
with the CPU generator hook, for ragged target counts incl. an image without targets, one target per image (100 groups) and 100
targets (one group)."""
import pytest
import torch

from custom_d_fine_amd import kernels
from custom_d_fine_amd.d_fine.arch import utils as U

pytestmark = pytest.mark.gpu


def _targets(counts, num_classes, device, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in counts:
        cxcy = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
        out.append({"labels": torch.randint(0, num_classes, (n,), generator=g).to(device), "boxes": torch.cat([cxcy, wh], 1).to(device)})
    return out


@pytest.mark.parametrize("counts", [[7, 3, 0, 12], [1, 1], [100, 37, 64], [5], [2, 0, 0, 9, 31, 4, 4, 17]])
@pytest.mark.parametrize("cpu_generator", [False, True])
def test_cdn_group_kernel_is_bit_identical_to_the_torch_composition(cuda, counts, cpu_generator):
    num_classes, num_queries = 80, 300
    emb = torch.nn.Embedding(num_classes + 1, 256, padding_idx=num_classes).to(cuda)
    targets = _targets(counts, num_classes, cuda, seed=sum(counts))
    outs = []
    try:
        for use_kernel in (True, False):
            kernels.CDN_KERNEL = use_kernel
            if cpu_generator:
                U.set_denoising_generator(torch.Generator().manual_seed(11))
            else:
                torch.manual_seed(123)
            with torch.no_grad():
                outs.append(U.get_contrastive_denoising_training_group(targets, num_classes, num_queries, emb, num_denoising=100,
                                                                       label_noise_ratio=0.5, box_noise_scale=1.0))
    finally:
        kernels.CDN_KERNEL = True
        U.set_denoising_generator(None)
    (la, ba, ma, meta_a), (lb, bb, mb, meta_b) = outs
    assert la.shape == lb.shape and torch.equal(la, lb)                       # same class ids -> same embedding rows
    assert ba.dtype == torch.float32 and ba.shape == bb.shape
    assert torch.equal(ba.view(torch.int32), bb.view(torch.int32)), (ba - bb).abs().max().item()
    assert torch.equal(ma, mb) and ma.dtype == torch.bool
    assert meta_a["dn_num_group"] == meta_b["dn_num_group"] and meta_a["dn_num_split"] == meta_b["dn_num_split"]
    assert all(torch.equal(x, y) for x, y in zip(meta_a["dn_positive_idx"], meta_b["dn_positive_idx"]))
    gmax = max(counts)
    assert ba.shape[1] == 2 * gmax * max(100 // gmax, 1)
    assert torch.isfinite(ba).all()


def test_cdn_attention_mask_is_shared_between_steps(cuda):
    """The mask depends on (total, num_queries, group size) only: one tensor per shape (its bit-packed forms are cached on its address)."""
    emb = torch.nn.Embedding(81, 256, padding_idx=80).to(cuda)
    t1, t2 = _targets([7, 3], 80, cuda, 1), _targets([2, 7], 80, cuda, 2)
    with torch.no_grad():
        m1 = U.get_contrastive_denoising_training_group(t1, 80, 300, emb)[2]
        m2 = U.get_contrastive_denoising_training_group(t2, 80, 300, emb)[2]
        m3 = U.get_contrastive_denoising_training_group(_targets([9, 3], 80, cuda, 3), 80, 300, emb)[2]
    assert m1.data_ptr() == m2.data_ptr() and m3.data_ptr() != m1.data_ptr()
    total = 2 * 7 * (100 // 7)
    ref = torch.zeros(total + 300, total + 300, dtype=torch.bool, device=cuda)
    ref[total:, :total] = True
    gid = torch.arange(total, device=cuda) // 14
    ref[:total, :total] = gid[:, None] != gid[None, :]
    assert torch.equal(m1, ref)
