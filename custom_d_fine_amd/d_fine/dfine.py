"""Model / loss / optimizer builders - the drop-in boundary of the hot path.

Same signatures as the reference's `src/d_fine/dfine.py:51-124`:
    build_model(model_name, num_classes, enable_mask_head, device, img_size=None, pretrained_model_path=None)
    build_loss(model_name, num_classes, label_smoothing, enable_mask_head)
    build_optimizer(model, lr, backbone_lr, betas, weight_decay, base_lr)
"""
from copy import deepcopy
from pathlib import Path

import torch.nn as nn
import torch.optim as optim

from .arch.dfine_decoder import DFINETransformer
from .arch.hgnetv2 import HGNetv2
from .arch.hybrid_encoder import HybridEncoder
from .configs import models
from .dfine_criterion import DFINECriterion
from .matcher import HungarianMatcher
from .utils import load_tuning_state

__all__ = ["DFINE", "build_model", "build_loss", "build_optimizer"]


class DFINE(nn.Module):
    __inject__ = ["backbone", "encoder", "decoder"]

    def __init__(self, backbone: nn.Module, encoder: nn.Module, decoder: nn.Module):
        super().__init__()
        self.backbone = backbone
        self.decoder = decoder
        self.encoder = encoder

    def forward(self, x, targets=None):
        return self.decoder(self.encoder(self.backbone(x)), targets)

    def deploy(self):
        self.eval()
        for m in self.modules():
            if hasattr(m, "convert_to_deploy"):
                m.convert_to_deploy()
        return self


def build_model(model_name, num_classes, enable_mask_head, device, img_size=None,
                pretrained_model_path=None):
    cfg = deepcopy(models[model_name])
    cfg["HybridEncoder"]["eval_spatial_size"] = img_size
    cfg["DFINETransformer"]["eval_spatial_size"] = img_size
    cfg["DFINETransformer"]["enable_mask_head"] = enable_mask_head
    model = DFINE(HGNetv2(**cfg["HGNetv2"]), HybridEncoder(**cfg["HybridEncoder"]),
                  DFINETransformer(num_classes=num_classes, **cfg["DFINETransformer"]))
    if pretrained_model_path:
        if not Path(pretrained_model_path).exists():
            raise FileNotFoundError(f"{pretrained_model_path} does not exist")
        model = load_tuning_state(model, str(pretrained_model_path))
    return model.to(device)


def build_loss(model_name, num_classes, label_smoothing, enable_mask_head):
    cfg = deepcopy(models[model_name])
    if enable_mask_head and "masks" not in cfg["DFINECriterion"]["losses"]:
        # the reference appends to a list shared by all sizes (dfine.py:75-76), so a second call
        # would duplicate the mask loss; here the config is copied first.
        cfg["DFINECriterion"]["losses"].append("masks")
    matcher = HungarianMatcher(**cfg["matcher"])
    return DFINECriterion(matcher, num_classes=num_classes, label_smoothing=label_smoothing,
                          **cfg["DFINECriterion"])


def param_groups(model, backbone_lr, base_lr):
    """The reference's four AdamW groups by parameter-name substrings (dfine.py:87-122):
    backbone w/o norm | backbone norm (wd 0) | enc/dec norm+bias (wd 0) | rest."""
    g = [[], [], [], []]
    for name, p in model.named_parameters():
        is_norm = "norm" in name or "bn" in name
        if "backbone" in name:
            g[1 if is_norm else 0].append(p)
        elif ("encoder" in name or "decoder" in name) and (is_norm or "bias" in name):
            g[2].append(p)
        else:
            g[3].append(p)
    return [
        {"params": g[0], "lr": backbone_lr, "initial_lr": backbone_lr},
        {"params": g[1], "lr": backbone_lr, "weight_decay": 0.0, "initial_lr": backbone_lr},
        {"params": g[2], "weight_decay": 0.0, "lr": base_lr, "initial_lr": base_lr},
        {"params": g[3], "lr": base_lr, "initial_lr": base_lr},
    ]


def build_optimizer(model, lr, backbone_lr, betas, weight_decay, base_lr):
    return optim.AdamW(param_groups(model, backbone_lr, base_lr), lr=lr, betas=betas,
                       weight_decay=weight_decay)
