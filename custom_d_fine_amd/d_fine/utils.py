"""Checkpoint loading for fine-tuning (ref `src/d_fine/utils.py:156-181`).

Upstream D-FINE checkpoints trained on Objects365 (366 classes) are remapped onto COCO's 80
classes through the id table below; tensors whose shape still does not match are skipped.
"""
from typing import Dict

import torch

from .dist_utils import is_main_process

# COCO class k  <->  Objects365 class obj365_ids[k]  (rows are shifted by one in the checkpoints)
obj365_ids = [
    0, 46, 5, 58, 114, 55, 116, 65, 21, 40, 176, 127, 249, 24, 56, 139, 92, 78, 99, 96,
    144, 295, 178, 180, 38, 39, 13, 43, 120, 219, 148, 173, 165, 154, 137, 113, 145, 146, 204, 8,
    35, 10, 88, 84, 93, 26, 112, 82, 265, 104, 141, 152, 234, 143, 150, 97, 2, 50, 25, 75,
    98, 153, 37, 73, 115, 132, 106, 61, 163, 134, 277, 81, 133, 18, 94, 30, 169, 70, 328, 226,
]


def map_class_weights(cur_tensor, pretrain_tensor):
    if pretrain_tensor.size() == cur_tensor.size():
        return pretrain_tensor
    out = cur_tensor.clone()
    out.requires_grad = False
    coco = torch.arange(len(obj365_ids))
    obj = torch.tensor(obj365_ids) + 1
    if pretrain_tensor.size() > cur_tensor.size():
        out[coco] = pretrain_tensor[obj]
    else:
        out[obj] = pretrain_tensor[coco]
    return out


def adjust_head_parameters(cur_state_dict, pretrain_state_dict):
    key = "decoder.denoising_class_embed.weight"
    if pretrain_state_dict[key].size() != cur_state_dict[key].size():
        del pretrain_state_dict[key]
    names = ["decoder.enc_score_head.weight", "decoder.enc_score_head.bias"]
    for i in range(8):
        names += [f"decoder.dec_score_head.{i}.weight", f"decoder.dec_score_head.{i}.bias"]
    for n in names:
        if n in cur_state_dict and n in pretrain_state_dict:
            pretrain_state_dict[n] = map_class_weights(cur_state_dict[n], pretrain_state_dict[n])
    return pretrain_state_dict


def matched_state(state: Dict[str, torch.Tensor], params: Dict[str, torch.Tensor]):
    hit, missed, unmatched = {}, [], []
    for k, v in state.items():
        if k not in params:
            missed.append(k)
        elif v.shape != params[k].shape:
            unmatched.append(k)
        else:
            hit[k] = params[k]
    return hit, {"missed": missed, "unmatched": unmatched}


def load_tuning_state(model, path: str):
    if path.startswith("http"):
        raise RuntimeError("no network on the target machines: download the checkpoint first")
    state = torch.load(path, map_location="cpu", weights_only=True)
    if "ema" in state:
        state = state["ema"]["module"]
    elif "model" in state:
        state = state["model"]
    try:
        stat, infos = matched_state(model.state_dict(),
                                    adjust_head_parameters(model.state_dict(), state))
    except Exception:
        stat, infos = matched_state(model.state_dict(), state)
    model.load_state_dict(stat, strict=False)
    if is_main_process():
        print(f"Pretrained weights from {path}: {len(stat)} tensors loaded, "
              f"{len(infos['missed'])} missing, {len(infos['unmatched'])} shape-mismatched")
    return model
