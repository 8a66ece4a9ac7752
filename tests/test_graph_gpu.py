"""HIP-graph replay of backbone + encoder (dl/engine.GraphedSegment) against the eager launch sequence it records:
same features, same gradients in the fused optimizer's flat buffer, same BatchNorm statistics and counters, same training
trajectory - the graph changes who issues the launches, not what runs (reference step: src/dl/train.py:550-586)."""
import copy

import pytest
import torch

import bench
from custom_d_fine_amd.d_fine.arch import utils as U
from custom_d_fine_amd.dl.engine import GraphedSegment, _BackboneEncoder
from custom_d_fine_amd.dl.synthetic import make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_bn_counter_mode():
    """TrainStep switches the BatchNorm step counters to the deferred mode for the process; later test modules expect the
    immediate mode."""
    yield
    from custom_d_fine_amd import kernels
    kernels.flush_bn_counters()
    kernels.defer_bn_counters(False)


def _segment_grads_eager(step, images, gouts):
    """Eager backbone + encoder forward / backward with the given output gradients -> (features, flat gradient buffer)."""
    fused = step.fused
    be = _BackboneEncoder(step.model.backbone, step.model.encoder)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = be(images)
    torch.autograd.backward(feats, [g.clone() for g in gouts])       # (clones: DFINE_PARK_EAGER adds onto them in place)
    fused._collect_grads()
    fused._uses.clear()
    return [f.detach().clone() for f in feats], fused.flat_grad.clone()


@pytest.mark.parametrize("name,img,bs", [("n", 320, 4), ("m", 320, 2)])
def test_graphed_segment_equals_eager_segment(cuda, monkeypatch, name, img, bs):
    torch.manual_seed(0)
    step = bench.build_step(name, img, cuda, torch.bfloat16)
    fused = step.fused
    images, _ = make_batch(bs, img, seed=3, device=cuda)
    from custom_d_fine_amd import kernels
    kernels.defer_bn_counters(True)
    # the captured segment hands gradients of maps with several consumers from kernel to kernel where the eager pass lets
    # autograd add them (kernels.park_grad: the in-place adds are only safe on the segment's own buffers); the eager reference
    # of THIS test runs the same hand-offs on cloned output gradients, so both sides are the same launches
    monkeypatch.setenv("DFINE_PARK_EAGER", "1")
    kernels.reload_env()
    buf0 = fused.flat_buf.clone()
    be = _BackboneEncoder(step.model.backbone, step.model.encoder)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        shapes = [f.shape for f in be(images)]
    fused.flat_buf.copy_(buf0)
    kernels._BN_PENDING.clear()
    g = torch.Generator(device="cpu").manual_seed(5)
    gouts = [(torch.randn(s, generator=g) * 1e-2).to(cuda, torch.bfloat16) for s in shapes]

    feats_e, grad_e = _segment_grads_eager(step, images, gouts)
    buf_e = fused.flat_buf.clone()
    # run-to-run spread of the eager gradients themselves (float atomics in the depthwise / stem / BatchNorm reductions)
    fused.flat_grad.zero_()
    fused.flat_buf.copy_(buf0)
    kernels._BN_PENDING.clear()
    _, grad_e2 = _segment_grads_eager(step, images, gouts)
    noise = (grad_e2 - grad_e).abs().max().item()
    pend_e = {k: v[1] for k, v in kernels._BN_PENDING.items()}
    fused.flat_grad.zero_()
    fused.flat_buf.copy_(buf0)
    kernels._BN_PENDING.clear()

    seg = GraphedSegment(be, (images,), amp_dtype=torch.bfloat16, fused=fused)
    # building the segment (warm-up runs + capture) must leave no trace in the training state
    assert torch.equal(fused.flat_buf, buf0) and float(fused.flat_grad.abs().max()) == 0.0
    assert not kernels._BN_PENDING and not fused._deferred
    for rep in range(4):                                   # replays are repeatable
        fused.flat_grad.zero_()
        fused.flat_buf.copy_(buf0)
        kernels._BN_PENDING.clear()
        feats = seg(images)
        torch.autograd.backward(feats, gouts)
        torch.cuda.synchronize()
        for a, b in zip(feats, feats_e):
            assert torch.equal(a, b), "captured forward differs from the eager forward"
        assert torch.equal(fused.flat_buf, buf_e), "BatchNorm running statistics differ"
        assert {k: v[1] for k, v in kernels._BN_PENDING.items()} == pend_e
        delta = (fused.flat_grad - grad_e).abs()
        scale = grad_e.abs().max().item()
        names = {id(p): n for n, p in step.model.named_parameters()}
        worst = []
        for i, p in enumerate(fused._params):
            o, name = fused.grad_offset(i), names[id(p)]
            d, sc = delta[o:o + p.numel()].max().item(), grad_e[o:o + p.numel()].abs().max().item()
            # the stem's weight gradients are sums of ~1e5 same-sign products reduced by float atomics: their run-to-run
            # spread reaches 2e-3 of the value (one eager pair does not always show it); everything else is tight
            tol = 5e-3 * sc if "backbone.stem" in name and name.endswith("conv.weight") else max(8 * noise, 2e-4 * scale)
            if d > tol:
                worst.append((d, tol, sc, name))
        assert not worst, (rep, noise, scale, sorted(worst, reverse=True)[:5])
    kernels._BN_PENDING.clear()
    kernels.defer_bn_counters(False)
    monkeypatch.delenv("DFINE_PARK_EAGER")
    kernels.reload_env()


def _spy_grads(step, seen):
    """Records the flat gradient buffer right before every optimizer step."""
    orig, fused = step.fused.step, step.fused

    def spy():
        fused._collect_grads()
        torch.cuda.synchronize()
        seen.append(fused.flat_grad.clone())
        orig()
    fused.step = spy


def _rel(a, b):
    return ((a - b).norm() / a.norm()).item()


def test_graphed_train_steps_track_eager_steps(cuda):
    """Whole train steps, graph against eager from the same initial state: the first step's losses and gradients agree to
    the non-determinism of the decoder's atomics (measured on a second eager run), and so does the state after a few steps."""
    res = {}
    for graph in (False, "again", True):
        torch.manual_seed(0)
        step = bench.build_step("s", 320, cuda, torch.bfloat16)
        step.hip_graph = graph is True
        images, targets = make_batch(4, 320, seed=1, device=cuda)
        losses, grads = [], []
        _spy_grads(step, grads)
        for it in range(5):
            U.set_denoising_generator(torch.Generator().manual_seed(100 + it))
            loss, _ = step(images, targets)
            losses.append(loss.item())
        U.set_denoising_generator(None)
        if graph is True:
            assert step._graphs, "the graphed path did not run"
        nbt = [b.item() for n, b in step.model.named_buffers() if n.endswith("num_batches_tracked")]
        res[graph] = (losses, step.fused.flat_param.clone(), step.fused.flat_buf.clone(), nbt, grads)
    g_e, g_e2, g_g = res[False][4][0], res["again"][4][0], res[True][4][0]
    assert torch.isfinite(g_g).all()
    noise = _rel(g_e, g_e2)
    assert _rel(g_e, g_g) <= max(4 * noise, 1e-5), (_rel(g_e, g_g), noise)
    assert (g_e - g_g).abs().max().item() <= max(4 * (g_e - g_e2).abs().max().item(), 1e-6 * g_e.abs().max().item())
    le, lg = res[False][0], res[True][0]
    assert abs(le[0] - lg[0]) <= 2e-3 * abs(le[0]), (le, lg)
    # later steps: the trajectory amplifies the atomics' rounding noise of step 1 - two EAGER runs of this loop differ by up to 3 %
    # in the loss of step 5 (measured: 32.56 / 33.53 / 32.70 over three runs) - so only the eager runs' own spread is demanded
    spread = max(abs(a - b) / abs(a) for a, b in zip(le, res["again"][0]))
    assert max(abs(a - b) / abs(a) for a, b in zip(le, lg)) < max(3 * spread, 5e-2), (le, lg, spread)
    assert res[False][3] == res[True][3] and set(res[True][3]) == {5}
    pe, pg = res[False][1], res[True][1]
    cos = torch.nn.functional.cosine_similarity(pe - pe.mean(), pg - pg.mean(), dim=0).item()
    assert cos > 0.9999, cos


def test_graphed_steps_with_gradient_accumulation(cuda):
    """Two micro-steps per optimizer step: the backward graph ADDS into the flat gradient buffer."""
    out = {}
    for graph in (False, "again", True):
        torch.manual_seed(0)
        step = bench.build_step("n", 320, cuda, torch.bfloat16)
        step.hip_graph, step.accum_steps = graph is True, 2
        batches = [make_batch(2, 320, seed=s, device=cuda) for s in (1, 2)]
        seen = []
        _spy_grads(step, seen)
        for it in range(4):
            U.set_denoising_generator(torch.Generator().manual_seed(7 + it))
            step(*batches[it % 2])
        U.set_denoising_generator(None)
        out[graph] = seen
    assert len(out[True]) == len(out[False]) == 2
    a, b = out[False][0], out[True][0]
    rel, noise = _rel(a, b), _rel(a, out["again"][0])        # (two EAGER runs differ by the decoder's atomics: usually < 1 %, seen > 2 %)
    assert torch.isfinite(b).all() and rel < max(2e-2, 3 * noise), (rel, noise)
    assert abs(a.norm().item() / b.norm().item() - 1) < 1e-2


def test_encoder_buckets_start_before_backbone_backward(cuda):
    """Graph mode, data-parallel bookkeeping on (overlap=True): backbone and encoder are separate graphed segments, so the encoder's
    gradient buckets are complete - gathered, their deferred partial sums reduced, their all-reduce started - BEFORE the backbone's
    backward graphs are replayed (one segment delivered everything at the very end of backward).  Reference: DistributedDataParallel's
    overlapped bucket all-reduce, /root/reference/src/dl/train.py:171-176."""
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep
    from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
    torch.manual_seed(0)
    model = dfine.build_model("n", 5, False, str(cuda), img_size=[320, 320]).train()
    crit = dfine.build_loss("n", 5, 0.0, False)
    ema = ModelEMA(model, 0.9998)
    opt = dfine.build_optimizer(model, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    fused = FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1, overlap=True, bucket_mb=2)
    step = TrainStep(model, crit, opt, amp_dtype=torch.bfloat16, clip_max_norm=0.1, ema=ema, fused_optimizer=fused, hip_graph=True)
    images, targets = make_batch(2, 320, num_classes=5, seed=3, device=cuda)
    for _ in range(2):
        step(images, targets)
    assert step._graphs
    chain = next(iter(step._graphs.values()))
    bb, enc = chain.segments
    log, state = [], []

    class _Spy:
        def __init__(self, g):
            self.g = g

        def replay(self):
            log.append("backbone_backward")
            state.extend((b["owner"], b["done"]) for b in fused._buckets)     # done = its all-reduce has been started
            self.g.replay()
    gm, gs = bb.bwd_pairs[0]
    bb.bwd_pairs[0] = (_Spy(gm), gs)
    loss, _ = step(images, targets)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert log == ["backbone_backward"] and state
    owners = [o for o, _ in state]
    # launch order = the order the modules' backward passes end in
    assert owners == sorted(owners, key=("decoder", "encoder", "backbone").index), owners
    assert all(done for o, done in state if o in ("decoder", "encoder")), state     # ... even with parameters that got no gradient
    assert not any(done for o, done in state if o == "backbone"), state
