"""Shared norm layer: batch-norm with frozen statistics (l / x backbones).

Buffer names (weight, bias, running_mean, running_var) match the reference's
`src/d_fine/arch/common.py:29-70` so checkpoints load unchanged.
"""
import torch
import torch.nn as nn


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def affine(self):
        """Per-channel (scale, shift) so that y = x*scale + shift."""
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.affine()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}"


def freeze_batch_norm2d(module: nn.Module) -> nn.Module:
    """Recursively swaps nn.BatchNorm2d for FrozenBatchNorm2d (fresh unit statistics, like
    the reference's `HGNetv2._freeze_norm`, hgnetv2.py:547-555)."""
    if isinstance(module, nn.BatchNorm2d):
        return FrozenBatchNorm2d(module.num_features)
    for name, child in module.named_children():
        new = freeze_batch_norm2d(child)
        if new is not child:
            setattr(module, name, new)
    return module
