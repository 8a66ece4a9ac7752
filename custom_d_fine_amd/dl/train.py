"""`python -m custom_d_fine_amd.dl.train [key=value ...]` (alias: `python -m src.dl.train`).

Train-loop counterpart of the reference's `src/dl/train.py` for the hot path only: model / loss /
optimizer construction, data-parallel set-up, AMP, the optimisation step, `last.pt` / `model.pt`
weight-only checkpoints.  Dataset loading, evaluation and logging
are outside the path; a synthetic COCO-shaped loader stands in for the data pipeline.
Configuration: the reference's YAML keys (`config.yaml`) given as `key=value` overrides or a YAML
file via `config=path.yaml`; no hydra.
"""
import os
import sys
import time
from pathlib import Path

import torch

from ..d_fine import dist_utils
from ..d_fine.dfine import build_loss, build_model, build_optimizer
from .engine import ModelEMA, TrainStep, wrap_data_parallel
from .synthetic import make_batch

DEFAULTS = {
    "model_name": "m", "task": "detect", "exp": "exp",
    "train": {
        "device": "cuda", "num_classes": 80, "img_size": [640, 640], "batch_size": 32, "b_accum_steps": 1,
        "epochs": 1, "steps_per_epoch": 50, "amp_enabled": True, "amp_dtype": "bf16", "clip_max_norm": 0.1,
        "use_ema": True, "ema_momentum": 0.9998, "base_lr": 1.5e-4, "backbone_lr": 2e-5,
        "betas": [0.9, 0.999], "weight_decay": 1.25e-4, "cycler_pct_start": 0.1, "use_scheduler": True,
        "label_smoothing": 0.0, "seed": 42, "ddp": {"enabled": False}, "path_to_save": "output/models/exp",
        "pretrained_model_path": None, "resume_path": None,
        "data_path": None,           # a folder with images/ + labels/ (YOLO txt): dl/data_device.YoloTxtDataset; None = synthetic batches
        "multiscale_prob": 0.0,      # the reference's train_collate_fn resize (dataset.py:667-694), on the device
        "mosaic_prob": 0.0,          # 2 x 2 mosaic + random affine per sample (dataset.py:258-345,386-392), on the device
    },
}


def _merge(cfg, key, value):
    parts = key.split(".")
    d = cfg
    for p in parts[:-1]:
        d = d.setdefault(p, {})
    import yaml
    d[parts[-1]] = yaml.safe_load(value)


def load_config(argv):
    import copy
    import yaml
    cfg = copy.deepcopy(DEFAULTS)
    for a in argv:
        k, _, v = a.partition("=")
        if k == "config":
            user = yaml.safe_load(open(v))

            def rec(dst, src):
                for kk, vv in src.items():
                    if isinstance(vv, dict) and isinstance(dst.get(kk), dict):
                        rec(dst[kk], vv)
                    else:
                        dst[kk] = vv
            rec(cfg, user)
    for a in argv:
        k, _, v = a.partition("=")
        if k != "config":
            _merge(cfg, k, v)
    return cfg


def shard_batches(n, rank, world, batch_size, seed):
    """Index lists of one rank's batches for one epoch over a dataset of `n` samples: torch's DistributedSampler with
    shuffle and drop_last=False (ref dataset.py:562-568) followed by a DataLoader that keeps the short last batch - a
    permutation seeded by `seed` (the same on every rank), wrapped around to world * ceil(n / world) entries, rank r takes
    entries r, r + world, ...  Every rank gets ceil(ceil(n / world) / batch_size) batches."""
    import random
    if n <= 0:
        return []
    order = list(range(n))
    random.Random(seed).shuffle(order)
    per_rank = -(-n // world)
    total = per_rank * world
    while len(order) < total:
        order += order[: total - len(order)]
    mine = order[rank:total:world]
    return [mine[i: i + batch_size] for i in range(0, per_rank, batch_size)]


class Trainer:
    def __init__(self, cfg):
        t = cfg["train"]
        self.cfg = cfg
        self.distributed = bool(t["ddp"]["enabled"]) and dist_utils.is_dist_available_and_initialized()
        self.rank, self.world = dist_utils.get_rank(), dist_utils.get_world_size()
        if t["device"] == "cuda" and torch.cuda.is_available():
            self.device = torch.device("cuda", dist_utils.get_local_rank())
        else:
            self.device = torch.device("cpu")
        torch.manual_seed(t["seed"] + (self.rank if self.distributed else 0))
        mask = cfg["task"] == "segment"
        self.model = build_model(cfg["model_name"], t["num_classes"], mask, str(self.device),
                                 img_size=t["img_size"], pretrained_model_path=t["pretrained_model_path"]).train()
        self.loss_fn = build_loss(cfg["model_name"], t["num_classes"], t["label_smoothing"], mask)
        self.ema = ModelEMA(self.model, t["ema_momentum"]) if t["use_ema"] else None
        self.optimizer = build_optimizer(self.model, lr=t["base_lr"], backbone_lr=t["backbone_lr"],
                                         betas=tuple(t["betas"]), weight_decay=t["weight_decay"],
                                         base_lr=t["base_lr"])
        fused = None
        if self.device.type == "cuda":
            from .fused_optim import FusedAdamWEMA
            fused = FusedAdamWEMA(self.model, self.optimizer, self.ema, clip_max_norm=t["clip_max_norm"])
            fused.broadcast_from_rank0()
        elif self.distributed:
            self.model = wrap_data_parallel(self.model, self.device)
        if t.get("data_path"):           # one epoch = one pass over this rank's shard of the folder
            from . import data_device
            self._dataset = data_device.YoloTxtDataset(t["data_path"], t["img_size"])
            t["steps_per_epoch"] = len(shard_batches(len(self._dataset), self.rank, self.world, t["batch_size"], 0))
        sched = None
        if t["use_scheduler"]:
            max_lr = t["base_lr"] * 2
            if cfg["model_name"] in ("l", "x") or mask:       # per-group maxima for the big models (ref train.py:203-214)
                max_lr = [t["backbone_lr"] * 2, t["backbone_lr"] * 2, t["base_lr"] * 2, t["base_lr"] * 2]
            sched = torch.optim.lr_scheduler.OneCycleLR(
                self.optimizer, max_lr=max_lr, epochs=t["epochs"],
                steps_per_epoch=max(t["steps_per_epoch"] // max(t["b_accum_steps"], 1), 1),
                pct_start=t["cycler_pct_start"], cycle_momentum=False)
        amp = None
        if t["amp_enabled"] and self.device.type == "cuda":
            amp = torch.bfloat16 if t["amp_dtype"] == "bf16" else torch.float16
        self.step = TrainStep(self.model, self.loss_fn, self.optimizer, amp_dtype=amp,
                              clip_max_norm=t["clip_max_norm"], ema=self.ema, scheduler=sched,
                              accum_steps=t["b_accum_steps"], fused_optimizer=fused,
                              hip_graph=fused is not None and os.environ.get("DFINE_HIPGRAPH", "1") == "1")
        self.path_to_save = Path(t["path_to_save"])
        self.fused, self.scheduler, self.start_epoch = fused, sched, 1
        if t.get("resume_path"):
            self.load_resume_state(t["resume_path"])

    def save_model(self):
        """Weight-only checkpoints, EMA weights when enabled (reference train.py:476-503)."""
        m = self.ema.model if self.ema is not None else self.model
        m = m.module if hasattr(m, "module") else m
        self.path_to_save.mkdir(parents=True, exist_ok=True)
        torch.save(m.state_dict(), self.path_to_save / "last.pt")
        torch.save(m.state_dict(), self.path_to_save / "model.pt")

    def save_resume_state(self, epoch):
        """`resume.pt`: everything a continued run needs - student and EMA weights, Adam moments and step counters (the
        fused optimizer keeps them in flat buffers), scheduler position, epoch.  (The reference only writes weight-only
        checkpoints, train.py:476-503, so an interrupted run restarts its schedule)"""
        m = self.model.module if hasattr(self.model, "module") else self.model
        state = {"epoch": epoch, "model": m.state_dict(),
                 "ema": None if self.ema is None else self.ema.model.state_dict(),
                 "optimizer": self.fused.state_dict() if self.fused is not None else self.optimizer.state_dict(),
                 "fused": self.fused is not None,
                 "scheduler": None if self.scheduler is None else self.scheduler.state_dict(),
                 "iters": self.step.iters}
        torch.save(state, self.path_to_save / "resume.pt")

    def load_resume_state(self, path):
        state = torch.load(path, map_location=self.device, weights_only=True)     # tensors and primitives only
        m = self.model.module if hasattr(self.model, "module") else self.model
        m.load_state_dict(state["model"])
        if self.ema is not None and state["ema"] is not None:
            self.ema.model.load_state_dict(state["ema"])
        if state["fused"] != (self.fused is not None):
            raise ValueError("resume.pt was written with a different optimizer path (fused HIP vs torch.optim)")
        (self.fused if self.fused is not None else self.optimizer).load_state_dict(state["optimizer"])
        if self.scheduler is not None and state["scheduler"] is not None:
            self.scheduler.load_state_dict(state["scheduler"])
            # OneCycleLR.load_state_dict restores its counters but not the rates it had written into the parameter groups,
            # and the fused optimizer's state holds no lr: without this the first resumed step runs at max_lr / 25
            for g, lr in zip(self.optimizer.param_groups, self.scheduler.get_last_lr()):
                g["lr"] = lr
        self.step.iters = state["iters"]
        self.start_epoch = state["epoch"] + 1
        from .. import kernels
        kernels.bump_weight_epoch()          # the weights changed under the cached packed / bf16 copies (and under a captured segment)

    @torch.no_grad()
    def evaluate(self, n_batches=2, conf_thresh=0.5, iou_thresh=0.5, keep_ratio=False):
        """Evaluation pass of the (EMA) model on synthetic batches: forward -> `preds_postprocess` / `gt_postprocess`
        (device-side top-K and box mapping) -> `Validator` metrics - the hand-off of the reference's
        `Trainer.get_preds_and_gt` + `evaluate` (train.py:384-474)."""
        from .postprocess import gt_postprocess, preds_postprocess
        from .validator import Validator
        t = self.cfg["train"]
        model = self.ema.model if self.ema is not None else self.model
        model = model.module if hasattr(model, "module") else model
        was_training = model.training
        model.eval()
        size = t["img_size"][0]
        all_preds, all_gt = [], []
        amp = self.step.amp_dtype
        for it in range(n_batches):
            images, targets = make_batch(t["batch_size"], size, t["num_classes"], seed=t["seed"] + 777 + it, device=self.device)
            orig = torch.tensor([[size, size]] * len(targets))
            if amp is not None:
                with torch.autocast(self.device.type, dtype=amp):
                    out = model(images)
            else:
                out = model(images)
            out = {k: v.float() for k, v in out.items() if k in ("pred_logits", "pred_boxes")}
            all_preds += preds_postprocess(images, out, orig, t["num_classes"], keep_ratio, conf_thresh)
            all_gt += gt_postprocess(images, targets, orig, keep_ratio)
        model.train(was_training)
        names = {i: str(i) for i in range(t["num_classes"])}
        return Validator(all_gt, all_preds, names, conf_thresh=conf_thresh, iou_thresh=iou_thresh).compute_metrics()

    def _batches(self, epoch):
        """(images, targets) of one epoch: a YOLO-txt folder sharded over the ranks like the reference's DistributedSampler
        (dataset.py:562-568: a per-epoch permutation shared by the ranks, padded by wrap-around to a multiple of the world
        size, every world-th index from `rank`; the loader keeps the short last batch), or synthetic batches.  Every rank
        yields the same number of batches (the fused optimizer all-reduces every step)."""
        import random
        t = self.cfg["train"]
        size = t["img_size"][0]
        if t.get("data_path"):
            from . import data_device
            if getattr(self, "_dataset", None) is None:
                self._dataset = data_device.YoloTxtDataset(t["data_path"], t["img_size"])
            # two streams: the permutation is shared by the ranks; mosaic partners, affine draws and the multiscale offset are
            # this rank's own (the reference's loader workers draw independently per rank)
            aug = random.Random((t["seed"], epoch, self.rank).__repr__())
            for idx in shard_batches(len(self._dataset), self.rank, self.world, t["batch_size"], t["seed"] + 7919 * epoch):
                images, targets = self._dataset.batch(idx, self.device, t.get("mosaic_prob", 0.0), aug)
                if t.get("multiscale_prob", 0) and aug.random() < t["multiscale_prob"] and self.device.type == "cuda":
                    images, targets = data_device.multiscale_collate(images, targets, aug.choice([-2, -1, 1, 2]) * 32)
                yield images, targets
            return
        for it in range(t["steps_per_epoch"]):
            yield make_batch(t["batch_size"], size, t["num_classes"], seed=t["seed"] + self.rank + 1000 * it, device=self.device,
                             with_masks=self.cfg["task"] == "segment")

    def train(self):
        t = self.cfg["train"]
        for epoch in range(self.start_epoch, t["epochs"] + 1):
            t0, losses = time.time(), []
            for images, targets in self._batches(epoch):
                loss, _ = self.step(images, targets)
                losses.append(loss)
            mean = torch.stack(losses).mean().item()
            if self.rank == 0:
                n = len(losses) * t["batch_size"] * self.world
                print(f"epoch {epoch}: loss {mean:.4f}, {n / (time.time() - t0):.1f} img/s", flush=True)
                self.save_model()
                self.save_resume_state(epoch)
            dist_utils.synchronize()


def main(argv=None):
    cfg = load_config(sys.argv[1:] if argv is None else argv)
    if cfg["train"]["ddp"]["enabled"]:
        dist_utils.init_distributed_mode()
    try:
        Trainer(cfg).train()
    finally:
        dist_utils.cleanup_distributed()


if __name__ == "__main__":
    main()
