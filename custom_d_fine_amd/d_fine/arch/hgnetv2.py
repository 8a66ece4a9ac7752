"""HGNetv2 backbone (B0-B6) with the reference's module tree / state-dict keys
(`src/d_fine/arch/hgnetv2.py`), written as table-driven builders.

Every conv unit is conv -> BN -> [ReLU -> [learnable affine]]; the unit's forward goes through
`kernels.conv_bn_act`, which is where the fused HIP path plugs in.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels
from .common import FrozenBatchNorm2d, freeze_batch_norm2d

__all__ = ["HGNetv2"]


class LearnableAffineBlock(nn.Module):
    """y = scale * x + bias with scalar parameters (ref hgnetv2.py:25-32)."""

    def __init__(self, scale_value=1.0, bias_value=0.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([scale_value]))
        self.bias = nn.Parameter(torch.tensor([bias_value]))

    def forward(self, x):
        return self.scale * x + self.bias


class ConvBNAct(nn.Module):
    """conv(bias=False) -> BN -> ReLU? -> LAB?   (ref hgnetv2.py:35-80)."""

    def __init__(self, in_chs, out_chs, kernel_size, stride=1, groups=1, padding="",
                 use_act=True, use_lab=False):
        super().__init__()
        self.use_act, self.use_lab = use_act, use_lab
        if padding == "same":
            self.conv = nn.Sequential(
                nn.ZeroPad2d([0, 1, 0, 1]),
                nn.Conv2d(in_chs, out_chs, kernel_size, stride, groups=groups, bias=False),
            )
        else:
            self.conv = nn.Conv2d(in_chs, out_chs, kernel_size, stride,
                                  padding=(kernel_size - 1) // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(out_chs)
        self.act = nn.ReLU() if use_act else nn.Identity()
        self.lab = LearnableAffineBlock() if (use_act and use_lab) else nn.Identity()

    def forward(self, x, pad_br=False, fanin=None, fans=None, residual=None):
        """pad_br: the input stands for F.pad(x, (0, 1, 0, 1)) (StemBlock); the pad is applied inside.
        residual: added to the unit's output (HG_Block's residual connection: in the BatchNorm apply pass where the unit is fused).
        fanin / fans: gradient hand-offs of HG_Block (kernels.GradFanIn)."""
        if isinstance(self.conv, nn.Sequential):
            x = self.conv[0](torch.cat(list(x), dim=1) if isinstance(x, (list, tuple)) else x)
            conv = self.conv[1]
        else:
            conv = self.conv
        lab = self.lab if isinstance(self.lab, LearnableAffineBlock) else None
        return kernels.conv_bn_act(x, conv, self.bn, "relu" if self.use_act else None, lab, pad_br=pad_br, fanin=fanin, fans=fans,
                                   residual=residual)


class LightConvBNAct(nn.Module):
    """1x1 conv+BN then depthwise kxk conv+BN+ReLU(+LAB)  (ref hgnetv2.py:83-112)."""

    def __init__(self, in_chs, out_chs, kernel_size, groups=1, use_lab=False):
        super().__init__()
        self.conv1 = ConvBNAct(in_chs, out_chs, 1, use_act=False, use_lab=use_lab)
        self.conv2 = ConvBNAct(out_chs, out_chs, kernel_size, groups=out_chs, use_act=True,
                               use_lab=use_lab)

    def forward(self, x, fanin=None):
        return self.conv2(self.conv1(x, fanin=fanin))


class StemBlock(nn.Module):
    """3x3/s2 -> (2x2 -> 2x2) || maxpool2/s1 -> concat -> 3x3/s2 -> 1x1  (ref hgnetv2.py:115-166)."""

    def __init__(self, in_chs, mid_chs, out_chs, use_lab=False):
        super().__init__()
        self.stem1 = ConvBNAct(in_chs, mid_chs, 3, stride=2, use_lab=use_lab)
        self.stem2a = ConvBNAct(mid_chs, mid_chs // 2, 2, stride=1, use_lab=use_lab)
        self.stem2b = ConvBNAct(mid_chs // 2, mid_chs, 2, stride=1, use_lab=use_lab)
        self.stem3 = ConvBNAct(mid_chs * 2, mid_chs, 3, stride=2, use_lab=use_lab)
        self.stem4 = ConvBNAct(mid_chs, out_chs, 1, stride=1, use_lab=use_lab)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=1, ceil_mode=True)

    def forward(self, x):
        # the two F.pad(., (0,1,0,1)) of the reference are folded into their consumers: the 2x2 convs and
        # the max-pool read zeros past the bottom / right edge (HIP stem kernels; ATen composition on CPU)
        x = self.stem1(x)
        # x has two consumers.  The pool is created first, so its backward runs last: stem2a's data gradient is parked and the
        # pool's backward adds its own onto it (kernels.GradFanIn) - no element-wise add of two [B, mid, H/2, W/2] maps
        fan = kernels.GradFanIn()
        pooled = kernels.stem_pool(x, fanin=fan)
        branch = self.stem2b(self.stem2a(kernels.park_grad(x, fan, owned=True), pad_br=True), pad_br=True)
        # the concatenation is never built on the GPU: stem3 reads both tensors in place (kernels._StemConv2)
        return self.stem4(self.stem3([pooled, branch]))


class EseModule(nn.Module):
    """Effective squeeze-excitation: x * sigmoid(conv1x1(mean_hw(x)))  (ref hgnetv2.py:169-186)."""

    def __init__(self, chs):
        super().__init__()
        self.conv = nn.Conv2d(chs, chs, kernel_size=1, stride=1, padding=0)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return x * self.sigmoid(self.conv(x.mean((2, 3), keepdim=True)))


class HG_Block(nn.Module):
    """layer_num conv units, dense concat of all intermediate maps, 1x1 aggregation
    ("se": squeeze+excite 1x1 pair; "ese": 1x1 + ESE), optional residual
    (ref hgnetv2.py:189-275)."""

    def __init__(self, in_chs, mid_chs, out_chs, layer_num, kernel_size=3, residual=False,
                 light_block=False, use_lab=False, agg="ese", drop_path=0.0):
        super().__init__()
        self.residual = residual
        unit = LightConvBNAct if light_block else ConvBNAct
        self.layers = nn.ModuleList()
        for i in range(layer_num):
            cin = in_chs if i == 0 else mid_chs
            if light_block:
                self.layers.append(unit(cin, mid_chs, kernel_size=kernel_size, use_lab=use_lab))
            else:
                self.layers.append(unit(cin, mid_chs, kernel_size=kernel_size, stride=1,
                                        use_lab=use_lab))
        total = in_chs + layer_num * mid_chs
        if agg == "se":
            self.aggregation = nn.Sequential(
                ConvBNAct(total, out_chs // 2, 1, stride=1, use_lab=use_lab),
                ConvBNAct(out_chs // 2, out_chs, 1, stride=1, use_lab=use_lab),
            )
        else:
            self.aggregation = nn.Sequential(
                ConvBNAct(total, out_chs, 1, stride=1, use_lab=use_lab), EseModule(out_chs))
        self.drop_path = nn.Dropout(drop_path) if drop_path else nn.Identity()

    def forward(self, x):
        feats = [x]
        if kernels.grad_fanin_enabled(x):
            # every map but the last has two consumers (the next layer and the aggregation): their data gradients meet in
            # the next layer's convolution epilogue instead of an element-wise add (kernels.GradFanIn)
            fans = [kernels.GradFanIn() for _ in self.layers]
            for layer, fan in zip(self.layers, fans):
                feats.append(layer(feats[-1], fanin=fan))
            if isinstance(self.aggregation[1], ConvBNAct):       # squeeze -> excitation: one consumer
                mid = self.aggregation[0](feats, fans=fans + [None])
                res = None
                if self.residual and isinstance(self.drop_path, nn.Identity):
                    # the residual connection rides in the excitation unit's BatchNorm apply pass; its gradient (the block
                    # output's) is parked for the aggregation's and layer 0's data gradients to add onto (captured segments).
                    # (made AFTER the squeeze unit: the parking node must run its backward before that unit's)
                    res = kernels.park_grad(x, fans[0]) if kernels.fanin_outer_enabled() else x
                y = self.aggregation[1](mid, residual=res)
                if res is not None:
                    return y
            else:
                y = self.aggregation[1](self.aggregation[0](feats, fans=fans + [None]))
        else:
            for layer in self.layers:
                feats.append(layer(feats[-1]))
            y = self.aggregation(feats)    # channel concat of all maps, read in place by the 1x1 aggregation conv
            fans = None
        if not self.residual:
            return y
        # x has a third consumer, the residual connection: its gradient (the block output's) is parked first and the aggregation's
        # and layer 0's data gradients are added onto it in place (captured segments only: see kernels.park_grad)
        return self.drop_path(y) + (kernels.park_grad(x, fans[0]) if (fans is not None and kernels.fanin_outer_enabled()) else x)


class HG_Stage(nn.Module):
    """Optional depthwise 3x3/s2 downsample + block_num HG_Blocks (ref hgnetv2.py:278-329)."""

    def __init__(self, in_chs, mid_chs, out_chs, block_num, layer_num, downsample=True,
                 light_block=False, kernel_size=3, use_lab=False, agg="se", drop_path=0.0):
        super().__init__()
        if downsample:
            self.downsample = ConvBNAct(in_chs, in_chs, 3, stride=2, groups=in_chs,
                                        use_act=False, use_lab=use_lab)
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[
            HG_Block(in_chs if i == 0 else out_chs, mid_chs, out_chs, layer_num,
                     residual=i > 0, kernel_size=kernel_size, light_block=light_block,
                     use_lab=use_lab, agg=agg,
                     drop_path=drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path)
            for i in range(block_num)
        ])

    def forward(self, x, fanin=None):
        """fanin: kernels.GradFanIn of x when x also leaves the backbone (HGNetv2.forward)."""
        if fanin is not None and isinstance(self.downsample, ConvBNAct):
            return self.blocks(self.downsample(x, fanin=fanin))
        return self.blocks(self.downsample(x))


_URL = "https://github.com/Peterande/storage/releases/download/dfinev1.0/PPHGNetV2_{}_stage1.pth"

# name: (stem [in, mid, out], 4 x stage [in, mid, out, blocks, downsample, light, k, layers])
_ARCH = {
    "B0": ([3, 16, 16], [[16, 16, 64, 1, False, False, 3, 3], [64, 32, 256, 1, True, False, 3, 3],
                         [256, 64, 512, 2, True, True, 5, 3], [512, 128, 1024, 1, True, True, 5, 3]]),
    "B1": ([3, 24, 32], [[32, 32, 64, 1, False, False, 3, 3], [64, 48, 256, 1, True, False, 3, 3],
                         [256, 96, 512, 2, True, True, 5, 3], [512, 192, 1024, 1, True, True, 5, 3]]),
    "B2": ([3, 24, 32], [[32, 32, 96, 1, False, False, 3, 4], [96, 64, 384, 1, True, False, 3, 4],
                         [384, 128, 768, 3, True, True, 5, 4], [768, 256, 1536, 1, True, True, 5, 4]]),
    "B3": ([3, 24, 32], [[32, 32, 128, 1, False, False, 3, 5], [128, 64, 512, 1, True, False, 3, 5],
                         [512, 128, 1024, 3, True, True, 5, 5], [1024, 256, 2048, 1, True, True, 5, 5]]),
    "B4": ([3, 32, 48], [[48, 48, 128, 1, False, False, 3, 6], [128, 96, 512, 1, True, False, 3, 6],
                         [512, 192, 1024, 3, True, True, 5, 6], [1024, 384, 2048, 1, True, True, 5, 6]]),
    "B5": ([3, 32, 64], [[64, 64, 128, 1, False, False, 3, 6], [128, 128, 512, 2, True, False, 3, 6],
                         [512, 256, 1024, 5, True, True, 5, 6], [1024, 512, 2048, 2, True, True, 5, 6]]),
    "B6": ([3, 48, 96], [[96, 96, 192, 2, False, False, 3, 6], [192, 192, 512, 3, True, False, 3, 6],
                         [512, 384, 1024, 6, True, True, 5, 6], [1024, 768, 2048, 3, True, True, 5, 6]]),
}


class HGNetv2(nn.Module):
    """Returns the stage outputs listed in `return_idx` (strides 4/8/16/32).

    Constructor signature as the reference (hgnetv2.py:424-434).  `pretrained=True` only
    loads `local_model_dir/PPHGNetV2_<name>_stage1.pth` (no network on the target machines).
    """

    arch_configs = {
        k: {"stem_channels": s, "stage_config": {f"stage{i + 1}": c for i, c in enumerate(st)},
            "url": _URL.format(k)}
        for k, (s, st) in _ARCH.items()
    }

    def __init__(self, name, use_lab=False, return_idx=[1, 2, 3], freeze_stem_only=True,
                 freeze_at=0, freeze_norm=True, pretrained=True,
                 local_model_dir="weight/hgnetv2/"):
        super().__init__()
        self.use_lab, self.return_idx = use_lab, return_idx
        stem, stages = _ARCH[name]
        self._out_strides = [4, 8, 16, 32]
        self._out_channels = [c[2] for c in stages]
        self.stem = StemBlock(stem[0], stem[1], stem[2], use_lab=use_lab)
        self.stages = nn.ModuleList(
            HG_Stage(cin, mid, cout, nblk, nlayer, down, light, k, use_lab)
            for cin, mid, cout, nblk, down, light, k, nlayer in stages)

        if freeze_at >= 0:
            self._freeze_parameters(self.stem)
            if not freeze_stem_only:
                for i in range(min(freeze_at + 1, len(self.stages))):
                    self._freeze_parameters(self.stages[i])
        if freeze_norm:
            freeze_batch_norm2d(self)
        if pretrained:
            path = os.path.join(local_model_dir, f"PPHGNetV2_{name}_stage1.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(
                    f"{path} not found; download {self.arch_configs[name]['url']} into "
                    f"{local_model_dir} (this build never touches the network)")
            self.load_state_dict(torch.load(path, map_location="cpu"))

    def _freeze_norm(self, m: nn.Module):
        return freeze_batch_norm2d(m)

    @staticmethod
    def _freeze_parameters(m: nn.Module):
        for p in m.parameters():
            p.requires_grad = False

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for i, stage in enumerate(self.stages):
            # a returned map that the next stage reads too has two consumers: the gradient arriving from outside is parked and
            # the stage's depthwise stride-2 convolution (created first = backward last) adds its data gradient onto it in place
            # (captured segments only, see kernels.park_grad)
            fan = kernels.GradFanIn() if (outs and outs[-1] is x and kernels.grad_fanin_enabled(x) and kernels.fanin_outer_enabled()) else None
            y = stage(x, fanin=fan) if fan is not None else stage(x)
            if fan is not None:
                outs[-1] = kernels.park_grad(x, fan)
            x = y
            if i in self.return_idx:
                outs.append(x)
        return outs
