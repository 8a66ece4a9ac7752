"""Multi-process path (world_size 2, gloo, CPU): data-parallel train step through DDP, the criterion's
folded all-reduce of the box-count normalisers and the dist_utils helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import torch_backend
    torch_backend.install()
    from custom_d_fine_amd.d_fine import dfine, dist_utils
    from custom_d_fine_amd.dl.engine import ModelEMA, TrainStep, wrap_data_parallel
    from custom_d_fine_amd.dl.synthetic import make_batch
    from tests import helpers

    dist_utils.init_distributed_mode()
    assert dist_utils.get_world_size() == world and dist_utils.get_rank() == rank
    assert dist_utils.all_gather_object({"r": rank}) == [{"r": 0}, {"r": 1}]
    assert dist_utils.broadcast_scalar(3.5 if rank == 0 else -1.0) == 3.5
    assert dist_utils.host_all_reduce_sum([rank + 1, 2.0]) == [3.0, 4.0]
    red = dist_utils.reduce_dict({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0)})
    assert red["a"].item() == 0.5 and red["b"].item() == 2.0

    model = dfine.build_model("n", 5, False, "cpu", img_size=[320, 320])
    model.load_state_dict(helpers.seeded_state_dict(model.state_dict()))
    model.train()
    crit = dfine.build_loss("n", 5, 0.0, False)
    ema = ModelEMA(model, 0.9998)
    ddp = wrap_data_parallel(model, torch.device("cpu"))
    opt = dfine.build_optimizer(ddp, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
    step = TrainStep(ddp, crit, opt, clip_max_norm=0.1, ema=ema)
    images, targets = make_batch(1, 320, num_classes=5, seed=42 + rank)       # different data per rank
    torch.manual_seed(7 + rank)
    loss, loss_dict = step(images, targets)
    assert torch.isfinite(loss)
    # every rank must hold identical parameters after the averaged-gradient step
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    # normalisers were averaged over ranks: sum over ranks of T / world
    n_local = float(len(targets[0]["labels"]))
    t = torch.tensor([n_local])
    dist.all_reduce(t)
    torch.save({"num_boxes_mean": t.item() / world, "loss": loss.item(), "n_losses": len(loss_dict)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist_utils.synchronize()
    dist_utils.cleanup_distributed()


def test_two_rank_gloo_train_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["num_boxes_mean"] == r1["num_boxes_mean"] and r0["n_losses"] == r1["n_losses"] == 38
