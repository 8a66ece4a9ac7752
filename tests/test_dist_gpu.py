"""The fused data-parallel train step (flat-buffer all-reduce + HIP clip/AdamW/EMA kernels) with world_size 2.
The GPU test box has ONE MI355X, and RCCL refuses two ranks on one device, so both ranks share cuda:0 and the
collectives go through gloo (device tensors are staged through the host): same code path as `bench.py --gpus N`
(FusedAdamWEMA.broadcast_from_rank0 / step, the criterion's folded all-reduce), different transport."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap, graph=False, backend="gloo"):
    local = rank if backend == "nccl" else 0           # RCCL: one device per rank; gloo: both ranks share the box's single GPU
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep
    from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
    from custom_d_fine_amd.dl.synthetic import make_batch

    torch.manual_seed(100 + rank)                      # different initial weights: rank 0's must win
    model = dfine.build_model("n", 5, False, str(dev), img_size=[320, 320]).train()
    crit = dfine.build_loss("n", 5, 0.0, False)
    ema = ModelEMA(model, 0.9998)
    model.backbone.stem.stem1.conv.weight.requires_grad_(False)     # a frozen parameter (the l / x configs freeze the stem)
    opt = dfine.build_optimizer(model, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    fused = FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1, overlap=overlap, bucket_mb=2)
    assert fused.overlap == overlap and (not overlap or len(fused._buckets) > 4)
    fused.broadcast_from_rank0()
    # every rank now holds rank 0's WHOLE state (frozen parameters, integer buffers and the EMA copy included)
    for name, sd in (("model", model.state_dict()), ("ema", ema.model.state_dict())):
        for k, v in sd.items():
            mine = v.detach().contiguous() if backend == "nccl" else v.detach().cpu().contiguous()
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert all(torch.equal(both[0], b) for b in both[1:]), f"{name}.{k} differs between the ranks after broadcast_from_rank0"
    step = TrainStep(model, crit, opt, amp_dtype=torch.bfloat16, clip_max_norm=0.1, ema=ema, fused_optimizer=fused, hip_graph=graph)
    images, targets = make_batch(2, 320, num_classes=5, seed=42 + rank, device=dev)      # different data per rank
    # ---- the bucketed all-reduce launched from backward hooks vs the single-shot all-reduce after backward, on gradients
    # that are deterministic functions of (rank, parameter): the reduced flat buffers must be bit-identical (the sum over
    # two ranks is order-independent; the buckets only change WHEN it runs).  A real backward is not comparable run to run:
    # the deformable-attention / BN kernels accumulate with atomics.
    import copy
    twin = copy.deepcopy(model)
    twin_opt = dfine.build_optimizer(twin, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    twin_fused = FusedAdamWEMA(twin, twin_opt, None, clip_max_norm=0.1, overlap=not overlap, bucket_mb=2)
    for f, mdl in ((fused, model), (twin_fused, twin)):
        gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
        fake = 0
        for p in mdl.parameters():
            if p.requires_grad:
                fake = fake + (p * torch.randn(p.shape, device=dev, generator=gen)).sum()
        fake.backward()
        f._collect_grads()
    torch.cuda.synchronize()
    assert torch.equal(fused.flat_grad, twin_fused.flat_grad), "bucketed and single-shot reductions differ"
    assert fused.flat_grad.abs().sum() > 0
    fused.flat_grad.zero_()
    twin_fused.flat_grad.zero_()
    # ---- the same comparison on a REAL backward (conv / linear weight gradients arrive through defer_wgrad, the learnable
    # affine scalars straight in the flat buffer, the rest through autograd hooks): no parameter may lose its gradient to
    # a bucket that was reduced too early.  Atomics make two backward passes differ in the last bits, hence tolerances.
    for f, mdl in ((fused, model), (twin_fused, twin)):
        torch.manual_seed(7 + rank)                     # same denoising noise for both
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mdl(images, targets=targets)
        with torch.autocast("cuda", enabled=False):
            ld = crit(out, targets)
        f.accumulating = False
        crit.total(ld).backward()
        f._collect_grads()
        f._uses.clear()
    torch.cuda.synchronize()
    ga, gb = fused.flat_grad, twin_fused.flat_grad
    assert torch.isfinite(ga).all() and torch.isfinite(gb).all()
    for i, p in enumerate(fused._params):
        o = fused.grad_offset(i)
        za, zb = bool((ga[o:o + p.numel()] != 0).any()), bool((gb[o:o + p.numel()] != 0).any())
        assert za == zb, f"parameter {i}: gradient present in one reduction mode only"
    assert (ga - gb).norm() <= 0.05 * gb.norm(), ((ga - gb).norm().item(), gb.norm().item())
    fused.flat_grad.zero_()
    del twin, twin_opt, twin_fused

    losses = []
    for _ in range(3 if graph else 2):
        loss, loss_dict = step(images, targets)
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert not graph or step._graphs, "the captured backbone + encoder segment did not run"
    flat = fused.flat_param.detach() if backend == "nccl" else fused.flat_param.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    flat = flat.cpu()
    assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged: the averaged-gradient step must keep them identical"
    assert all(torch.isfinite(torch.tensor(l)) for l in losses)
    torch.save({"losses": losses, "n_losses": len(loss_dict), "checksum": flat.double().sum().item(), "flat": flat},
               os.path.join(out_dir, f"rank{rank}_{int(overlap)}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,graph", [(True, False), (False, False), (True, True)])
def test_two_rank_fused_train_step_shared_gpu(cuda, tmp_path, overlap, graph):
    """graph: backbone + encoder through the captured HIP graphs (the bench's default mode) - the segment's gradients reach the
    flat buffer inside the graph, the buckets are reported after its replay, the GO count is all-reduced on the device."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), overlap, graph), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / f"rank0_{int(overlap)}.pt"), torch.load(tmp_path / f"rank1_{int(overlap)}.pt")
    assert r0["checksum"] == r1["checksum"] and r0["n_losses"] == r1["n_losses"]
    assert torch.equal(r0["flat"], r1["flat"])
    assert r0["losses"] != r1["losses"]                # different data per rank


def test_rccl_all_devices_fused_train_step(tmp_path):
    """The same step over RCCL with one rank per visible GPU (what `bench.py --gpus N` runs).  Needs at least two devices:
    skipped on the single-GPU test boxes - no multi-rank RCCL run exists for this build yet."""
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("one GPU visible: RCCL needs one device per rank")
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, True, "nccl"), nprocs=world, join=True)
    runs = [torch.load(tmp_path / f"rank{r}_1.pt") for r in range(world)]
    assert all(torch.equal(runs[0]["flat"], r["flat"]) for r in runs[1:])


def _worker_empty_rank(rank, world, port, out_dir, graph):
    """Rank 1's batch has NO targets: its criterion takes the host bookkeeping path while rank 0 builds its plans on the
    device.  Both must issue the same collectives (a mismatch pairs the GO-count all-reduce with a gradient bucket's)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import datetime
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from custom_d_fine_amd.d_fine import dfine
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep
    from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
    from custom_d_fine_amd.dl.synthetic import make_batch

    torch.manual_seed(100)
    model = dfine.build_model("n", 5, False, str(dev), img_size=[320, 320]).train()
    crit = dfine.build_loss("n", 5, 0.0, False)
    ema = ModelEMA(model, 0.9998)
    opt = dfine.build_optimizer(model, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    fused = FusedAdamWEMA(model, opt, ema, clip_max_norm=0.1, overlap=True, bucket_mb=2)
    fused.broadcast_from_rank0()
    step = TrainStep(model, crit, opt, amp_dtype=torch.bfloat16, clip_max_norm=0.1, ema=ema, fused_optimizer=fused, hip_graph=graph)
    images, targets = make_batch(2, 320, num_classes=5, seed=42 + rank, device=dev)
    if rank == 1:
        targets = [{**t, "labels": t["labels"][:0], "boxes": t["boxes"][:0]} for t in targets]
    losses = []
    for _ in range(3):
        loss, loss_dict = step(images, targets)
        losses.append(loss.item())
    torch.cuda.synchronize()
    flat = fused.flat_param.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged"
    assert all(torch.isfinite(torch.tensor(l)) for l in losses)
    torch.save({"losses": losses, "flat": flat}, os.path.join(out_dir, f"empty_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _spawn_bounded(fn, args, world, seconds):
    """mp.spawn with a wall-clock bound: a collective mismatch shows up as a hang, which must fail the test, not the box."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > seconds:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            pytest.fail(f"ranks did not finish within {seconds} s (mismatched collectives hang)")


@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_one_without_targets(cuda, tmp_path, graph):
    """the device-plan / host-plan choice is per rank; the collectives must not depend on it."""
    world = 2
    _spawn_bounded(_worker_empty_rank, (world, _free_port(), str(tmp_path), graph), world, 420)
    r0, r1 = torch.load(tmp_path / "empty_rank0.pt"), torch.load(tmp_path / "empty_rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])


def _worker_definition(rank, world, port, out_dir, overlap):
    """Data parallelism against its DEFINITION (reference train.py:171-176: the gradient every rank steps with is the mean
    of the ranks' own gradients; dfine_criterion.py:639-652: the normalisers are clamp(sum over ranks / world, 1)), not
    against another mode of the same code."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import copy
    import datetime
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    from custom_d_fine_amd.d_fine import dfine, dfine_criterion, dist_utils
    from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
    from custom_d_fine_amd.dl.synthetic import make_batch

    torch.manual_seed(100)
    model = dfine.build_model("n", 5, False, str(dev), img_size=[320, 320]).train()
    crit = dfine.build_loss("n", 5, 0.0, False)
    hp = dict(lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    opt = dfine.build_optimizer(model, **hp)
    fused = FusedAdamWEMA(model, opt, None, clip_max_norm=0.1, overlap=overlap, bucket_mb=2)
    fused.broadcast_from_rank0()

    def fake_backward(mdl):
        gen = torch.Generator(device="cuda").manual_seed(1234 + rank)          # a different gradient per rank
        fake = 0
        for p in mdl.parameters():
            if p.requires_grad:
                fake = fake + (p * torch.randn(p.shape, device=dev, generator=gen)).sum()
        fake.backward()

    # ---- each rank's OWN flat gradient, single-process: no collective is issued on this twin
    twin = copy.deepcopy(model)
    twin_fused = FusedAdamWEMA(twin, dfine.build_optimizer(twin, **hp), None, clip_max_norm=0.1, overlap=False)
    fake_backward(twin)
    twin_fused._gather(range(len(twin_fused._params)))
    twin_fused._flush_deferred()
    torch.cuda.synchronize()
    own = twin_fused.flat_grad.detach().cpu()
    every = [torch.zeros_like(own) for _ in range(world)]
    dist.all_gather(every, own)
    assert not torch.equal(every[0], every[1])
    expect_sum = every[0] + every[1]                     # two fp32 terms: the sum does not depend on the order
    # ---- the data-parallel reduction of the same gradients
    fake_backward(model)
    fused._collect_grads()
    torch.cuda.synchronize()
    assert torch.equal(fused.flat_grad.cpu(), expect_sum), "reduced flat gradient != g0 + g1"
    fused.flat_grad.zero_()
    # ---- ... and the step taken with it == clip(0.1) + AdamW on (g0 + g1) / 2, written with torch on the host
    before = fused.flat_param.detach().cpu().clone()
    mean = expect_sum / world                            # exact: a power of two
    ref_p, ref_o = [], []
    for (off, size), group in zip(fused.segments, opt.param_groups):
        t = before[off:off + size].clone().requires_grad_(True)
        t.grad = mean[off:off + size].clone()
        ref_p.append(t)
        ref_o.append(torch.optim.AdamW([t], lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"]))
    torch.nn.utils.clip_grad_norm_(ref_p, 0.1)
    for o in ref_o:
        o.step()
    fake_backward(model)
    fused.step()
    torch.cuda.synchronize()
    after = fused.flat_param.detach().cpu()
    for (off, size), t in zip(fused.segments, ref_p):
        got, ref = after[off:off + size], t.detach()
        assert (got - before[off:off + size]).abs().max() > 0
        assert torch.allclose(got, ref, rtol=1e-6, atol=3e-7), (got - ref).abs().max().item()

    # ---- the criterion's normalisers: the same outputs scored as rank `rank` of 2 and as a single process
    images, targets = make_batch(2, 320, num_classes=5, seed=42 + rank, device=dev)
    if rank == 1:
        targets = [{**t, "labels": t["labels"][:1], "boxes": t["boxes"][:1]} for t in targets]      # unequal target counts
    torch.manual_seed(7 + rank)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(images, targets=targets)
        with torch.autocast("cuda", enabled=False):
            ld2 = {k: float(v) for k, v in crit(out, targets).items()}
            norm2 = dict(crit._last_norm)
            saved = (dfine_criterion.get_world_size, dist_utils.get_world_size)
            dfine_criterion.get_world_size = dist_utils.get_world_size = lambda: 1
            try:
                ld1 = {k: float(v) for k, v in crit(out, targets).items()}
                norm1 = dict(crit._last_norm)
            finally:
                dfine_criterion.get_world_size, dist_utils.get_world_size = saved
    n_own = float(sum(len(t["labels"]) for t in targets))
    go_own = float(norm1["go_count"].item() if torch.is_tensor(norm1["go_count"]) else norm1["go_count"])
    counts = [None] * world
    dist.all_gather_object(counts, (n_own, go_own))
    nb = max((counts[0][0] + counts[1][0]) / world, 1.0)
    nb_go = max((counts[0][1] + counts[1][1]) / world, 1.0)
    assert counts[0][0] != counts[1][0]
    assert norm2["num_boxes"] == nb and norm1["num_boxes"] == max(n_own, 1.0)
    if norm2["num_boxes_go"] is not None:
        assert norm2["num_boxes_go"] == nb_go
    # a loss normalised by num_boxes (varifocal) / by the GO count (L1): world-2 value = world-1 value x own / averaged
    for key, own_n, avg_n in (("loss_vfl", max(n_own, 1.0), nb), ("loss_bbox", max(go_own, 1.0), nb_go),
                              ("loss_giou_aux_0", max(go_own, 1.0), nb_go), ("loss_vfl_pre", max(n_own, 1.0), nb)):
        assert ld1[key] > 0
        assert abs(ld2[key] - ld1[key] * own_n / avg_n) <= 2e-5 * abs(ld2[key]), (key, ld2[key], ld1[key], own_n, avg_n)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_reduction_is_the_mean_of_the_rank_gradients(cuda, tmp_path, overlap):
    """A17 against the definition: reduced flat gradient == g0 + g1 bit for bit (stepped with x 1/world), the parameters after
    the fused step == clip + AdamW on the mean gradient, the criterion's normalisers == clamp((T0 + T1) / 2, 1)."""
    _spawn_bounded(_worker_definition, (2, _free_port(), str(tmp_path), overlap), 2, 420)


def test_bench_dry_launcher_on_visible_devices(cuda):
    """`bench.py --gpus N --dry`: the launcher path (self-spawn through torch.distributed.run, RCCL group, one all-reduce) on
    however many devices this box has (N is capped at the visible devices)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    n = torch.cuda.device_count()
    assert line["dry"] is True and line["n_gpus"] == min(8, n) and line["all_reduce_of_ones"] == float(min(8, n))
