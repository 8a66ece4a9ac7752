"""A16 parity: flat-buffer clip + AdamW + EMA HIP kernels vs torch's clip_grad_norm_ + AdamW + the
reference's per-tensor EMA loop (plain PyTorch fp32 reference of the same op)."""
import copy
import math

import numpy as np

import pytest
import torch
import torch.nn as nn

from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.engine import ModelEMA
from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA

pytestmark = pytest.mark.gpu


# NB: no conv bias / affine shift directly in front of a BatchNorm (as in D-FINE): such a parameter has
# an exactly-zero gradient, Adam
# turns its float noise into +-lr updates and any two correct implementations diverge on it.
class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.encoder = nn.Sequential(nn.Conv2d(8, 8, 1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.decoder = nn.Linear(8, 5)
        self.register_buffer("anchors", torch.rand(1, 7, 4))

    def forward(self, x):
        return self.decoder(self.encoder(self.backbone(x)).mean((2, 3)))


def test_fused_step_matches_torch(cuda):
    torch.manual_seed(0)
    ref = Tiny().to(cuda)
    mine = copy.deepcopy(ref)
    ref_ema, mine_ema = ModelEMA(ref, 0.9998), ModelEMA(mine, 0.9998)
    kw = dict(lr=1e-3, backbone_lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-2, base_lr=1e-3)
    ref_opt, mine_opt = dfine.build_optimizer(ref, **kw), dfine.build_optimizer(mine, **kw)
    fused = FusedAdamWEMA(mine, mine_opt, mine_ema, clip_max_norm=0.1)
    sd_keys = list(mine.state_dict().keys())
    for it in range(1, 5):
        x = torch.randn(4, 3, 8, 8, device=cuda)
        for m in (ref, mine):
            m(x).square().sum().backward()
        for (n1, p1), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
            assert torch.allclose(p1.grad, p2.grad, rtol=1e-5, atol=1e-6), n1
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        ref_opt.step()
        ref_opt.zero_grad()
        # reference EMA loop (src/dl/train.py:52-73)
        mom = 0.9998 * (1 - math.exp(-it / 2000))
        with torch.no_grad():
            stu = ref.state_dict()
            for name, p in ref_ema.model.state_dict().items():
                if p.dtype.is_floating_point:
                    p *= mom
                    p += (1.0 - mom) * stu[name].detach()
        fused.step()
        assert fused.flat_grad.abs().max() == 0                             # zeroed for the next step
        ref_opt.param_groups[0]["lr"] *= 0.9           # scheduler-style lr change is picked up
        mine_opt.param_groups[0]["lr"] *= 0.9
        for (n1, p1), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
            assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7), (it, n1)
            assert p2.grad is None                                           # consumed by the step
        a, b = ref_ema.model.state_dict(), mine_ema.model.state_dict()
        for k in a:
            if a[k].dtype.is_floating_point:
                assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (it, k)
    assert list(mine.state_dict().keys()) == sd_keys
    # state-dict round trip still works on the flat views
    mine.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert torch.equal(mine.decoder.weight, ref.decoder.weight)
    assert mine.decoder.weight.data_ptr() >= fused.flat_param.data_ptr()


def test_fused_state_round_trip_resumes_identically(cuda):
    """model + EMA state dicts + FusedAdamWEMA.state_dict() are a complete checkpoint: a fresh instance that loads them
    continues with the same updates."""
    torch.manual_seed(1)
    kw = dict(lr=1e-3, backbone_lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-2, base_lr=1e-3)

    def make():
        m = Tiny().to(cuda)
        e = ModelEMA(m, 0.9998)
        o = dfine.build_optimizer(m, **kw)
        return m, e, o, FusedAdamWEMA(m, o, e, clip_max_norm=0.1)

    a, a_ema, _, a_fused = make()
    xs = [torch.randn(4, 3, 8, 8, device=cuda) for _ in range(5)]
    for x in xs[:3]:
        a(x).square().sum().backward()
        a_fused.step()
    ckpt = {"model": copy.deepcopy(a.state_dict()), "ema": copy.deepcopy(a_ema.model.state_dict()),
            "optim": a_fused.state_dict()}
    b, b_ema, _, b_fused = make()
    b.load_state_dict(ckpt["model"])
    b_ema.model.load_state_dict(ckpt["ema"])
    b_fused.load_state_dict(ckpt["optim"])
    assert b_fused.step_count == 3
    for x in xs[3:]:
        for m, f in ((a, a_fused), (b, b_fused)):
            m(x).square().sum().backward()
            f.step()
    # (equal up to the run-to-run noise of MIOpen's atomically accumulated conv weight gradients in this tiny torch model)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), n
    for (n, p), q in zip(a_ema.model.state_dict().items(), b_ema.model.state_dict().values()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-5, atol=1e-6), n
    with pytest.raises(ValueError):
        bad = a_fused.state_dict()
        bad["segments"] = [(0, 1)]
        b_fused.load_state_dict(bad)


def test_deferred_weight_gradients_match_immediate(cuda):
    """The conv / linear / attention backward ops hand split partial sums to the fused optimizer (one reduction launch per
    flush, straight into the flat gradient buffer) instead of returning gradient tensors: the flat gradients must be the
    ones the immediate path produces.  Backbone + hybrid encoder of D-FINE-m under bf16 autocast (1x1 / 3x3 / depthwise
    convs, channel-segment convs, Linear+activation, packed self-attention projections), twice per mode so that the
    accumulation into an already non-zero slot is covered too."""
    from custom_d_fine_amd.d_fine.dfine import build_model

    torch.manual_seed(3)
    model = build_model("m", 5, False, cuda, img_size=[320, 320]).train()
    for m in model.modules():                       # batch statistics would make run 2 differ from run 1 through the
        if isinstance(m, nn.BatchNorm2d):           # running buffers only in eval; in train mode they do not enter the output
            m.momentum = 0.0

    class BE(nn.Module):
        def __init__(self, full):
            super().__init__()
            self.backbone, self.encoder = full.backbone, full.encoder

        def forward(self, x):
            return self.encoder(self.backbone(x))

    net = BE(model)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    fused = FusedAdamWEMA(net, opt, None, clip_max_norm=0.1, overlap=False)
    x = torch.randn(4, 3, 320, 320, device=cuda)
    cot = None
    flats = {}
    for defer in (False, True, False):
        fused.defer_wgrads = defer
        fused.flat_grad.zero_()
        for rep in range(2):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats = net(x)
            if cot is None:
                cot = [torch.randn_like(f, dtype=torch.float32) / f.shape[1] for f in feats]
            sum((f.float() * c).sum() for f, c in zip(feats, cot)).backward()
            if defer:
                assert fused._deferred, "no backward op used the deferred path"
                assert sum(p.grad is None for p in net.parameters()) > 50
        fused._collect_grads()
        fused._uses.clear()
        assert not fused._deferred
        torch.cuda.synchronize()
        flats.setdefault(defer, []).append(fused.flat_grad.clone())
    ref, ref2, got = flats[False][0], flats[False][1], flats[True][0]
    assert torch.isfinite(got).all() and got.abs().sum() > 0
    noise = (ref - ref2).abs().max().item()                     # run-to-run noise of the immediate path itself (BN atomics)
    # (the deferred reduction itself is deterministic; what differs between two backward passes are the activations'
    # gradients that went through the BN / deformable-attention atomics, in bf16)
    tol = max(10 * noise, 2e-3 * ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol, ((got - ref).abs().max().item(), noise)
    # every parameter received its gradient through one of the two routes
    for i, p in enumerate(fused._params):
        o = fused._grad_offsets[i]
        assert (got[o:o + p.numel()] != 0).any() == (ref[o:o + p.numel()] != 0).any(), i


@pytest.mark.gpu
def test_side_stream_weight_gradients_match_single_stream(cuda, monkeypatch):
    """The deferred conv weight gradients (3x3, channel-segment 1x1, the grouped launches) and the depthwise / stem
    weight-gradient tensors run on a second HIP stream, joined in front of the optimizer's gather / reduction
    (hip._side_fork / side_join): the flat gradient must equal the single-stream one (same kernels; what differs run to run
    are activation gradients that went through atomics), over several steps so that a missing join would show."""
    from custom_d_fine_amd import hip
    from custom_d_fine_amd.d_fine.dfine import build_model

    torch.manual_seed(4)
    model = build_model("m", 5, False, cuda, img_size=[320, 320]).train()
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = 0.0

    class BE(nn.Module):
        def __init__(self, full):
            super().__init__()
            self.backbone, self.encoder = full.backbone, full.encoder

        def forward(self, x):
            return self.encoder(self.backbone(x))

    net = BE(model)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    fused = FusedAdamWEMA(net, opt, None, clip_max_norm=0.1, overlap=False)
    x = torch.randn(4, 3, 320, 320, device=cuda)
    cot, flats = None, {}
    for side in (False, True, False, True):
        monkeypatch.setattr(hip, "WGRAD_STREAM", side)
        fused.flat_grad.zero_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats = net(x)
        if cot is None:
            cot = [torch.randn_like(f, dtype=torch.float32) / f.shape[1] for f in feats]
        sum((f.float() * c).sum() for f, c in zip(feats, cot)).backward()
        if side:
            assert hip._SIDE_LIVE, "nothing was launched on the side stream"
        fused._collect_grads()
        fused._uses.clear()
        assert not hip._SIDE_LIVE and not fused._deferred
        for p in net.parameters():
            p.grad = None
        torch.cuda.synchronize()
        flats.setdefault(side, []).append(fused.flat_grad.clone())
    ref, ref2 = flats[False]
    noise = (ref - ref2).abs().max().item()
    tol = max(10 * noise, 2e-3 * ref.abs().max().item())
    for got in flats[True]:
        assert torch.isfinite(got).all() and got.abs().sum() > 0
        assert (got - ref).abs().max().item() <= tol, ((got - ref).abs().max().item(), noise)
        for i, p in enumerate(fused._params):
            o = fused._grad_offsets[i]
            assert (got[o:o + p.numel()] != 0).any() == (ref[o:o + p.numel()] != 0).any(), i


@pytest.mark.gpu
def test_deferred_reduction_with_repeated_module_and_accumulation(cuda):
    """A Linear applied FOUR times in one forward (the decoder's query_pos_head runs once per layer) over TWO micro-steps of a
    gradient-accumulation window leaves eight rows with one destination in the deferred table; they are reduced in
    stream-ordered rounds (a single launch would race on the read-modify-write).  The deferred flat gradient must equal the
    immediate one to fp32 rounding and be bit-identical from run to run."""
    from custom_d_fine_amd import kernels

    torch.manual_seed(7)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(64, 256), nn.Linear(256, 64)

        def forward(self, x):
            for _ in range(4):
                x = x + kernels.linear(kernels.linear(x, self.a.weight, self.a.bias, "relu"), self.b.weight, self.b.bias)
            return x

    net = Net().to(cuda)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    fused = FusedAdamWEMA(net, opt, None, clip_max_norm=0.1, overlap=False)
    xs = [torch.randn(8, 512, 64, device=cuda) * 0.5 for _ in range(2)]
    flats = {}
    for defer in (False, True, True):
        fused.defer_wgrads = defer
        fused.flat_grad.zero_()
        for x in xs:                                                 # two micro-steps, one flush
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            (y.float() ** 2).mean().backward()
        if defer:
            dsts = [d[1] for d in fused._deferred]
            assert len(dsts) == 32 and len(set(dsts)) == 4           # 4 parameters x 4 uses x 2 micro-steps
        fused._collect_grads()
        fused._uses.clear()
        assert not fused._deferred
        torch.cuda.synchronize()
        flats.setdefault(defer, []).append(fused.flat_grad.clone())
    ref, got, again = flats[False][0], flats[True][0], flats[True][1]
    assert torch.equal(got, again)
    assert got.abs().max() > 0
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-6 * ref.abs().max().item())


def test_bucket_all_reduces_start_in_one_fixed_order(cuda, monkeypatch):
    """RCCL pairs collectives by call order: every rank must issue the bucket all-reduces over the same flat-gradient ranges
    in the same order, whatever order its gradients arrive in and even when a parameter gets no gradient on this rank only
    (a batch without targets skips the denoising branch).  A recorder stands in for the collective; the world is told to have
    two ranks."""
    from custom_d_fine_amd.dl import fused_optim as fo

    torch.manual_seed(2)
    net = nn.Sequential(*[nn.Linear(64, 64) for _ in range(6)]).to(cuda)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    monkeypatch.setattr(fo, "get_world_size", lambda: 2)
    calls = []

    class _Work:
        def wait(self):
            pass

    def fake_all_reduce(t, async_op=False):
        calls.append((t.data_ptr(), t.numel()))
        return _Work()

    monkeypatch.setattr(fo.dist, "all_reduce", fake_all_reduce)
    fused = FusedAdamWEMA(net, opt, None, clip_max_norm=0.1, overlap=True, bucket_mb=64 * 65 * 4 * 2 / 2 ** 20)   # two layers per bucket
    assert len(fused._buckets) >= 3
    x = torch.randn(8, 64, device=cuda)

    def run(skip):
        calls.clear()
        h = x
        for i, layer in enumerate(net):
            if i != skip:
                h = layer(h)
        h.sum().backward()
        fused._collect_grads()
        for p in net.parameters():
            p.grad = None
        return list(calls)

    full = run(None)
    assert len(full) == len(fused._buckets) and len(set(full)) == len(full)
    for skip in (0, 2, 5):                     # a layer of the last / a middle / the first bucket gets no gradient
        assert run(skip) == full
