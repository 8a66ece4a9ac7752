"""HybridEncoder: 1x1 input projections, AIFI self-attention on the coarsest map, top-down
FPN + bottom-up PAN built from RepNCSPELAN4 blocks.

Module tree / parameter names follow the reference (`src/d_fine/arch/hybrid_encoder.py`) so
checkpoints are interchangeable; forward math goes through `custom_d_fine_amd.kernels`.
"""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels
from .utils import get_activation

__all__ = ["HybridEncoder"]


def _fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(kernel, bias) of the conv equivalent to bn(conv(x)) in eval mode."""
    std = (bn.running_var + bn.eps).sqrt()
    g = bn.weight / std
    return conv.weight * g.reshape(-1, 1, 1, 1), bn.bias - bn.running_mean * g


class ConvNormLayer_fuse(nn.Module):
    """conv -> BN -> act with `convert_to_deploy()` folding BN into the conv
    (ref hybrid_encoder.py:21-79)."""

    def __init__(self, ch_in, ch_out, kernel_size, stride, g=1, padding=None, bias=False, act=None):
        super().__init__()
        padding = (kernel_size - 1) // 2 if padding is None else padding
        self.conv = nn.Conv2d(ch_in, ch_out, kernel_size, stride, groups=g, padding=padding, bias=bias)
        self.norm = nn.BatchNorm2d(ch_out)
        self.act = nn.Identity() if act is None else get_activation(act)
        self._act_name = act if isinstance(act, str) else None
        self.ch_in, self.ch_out, self.kernel_size, self.stride = ch_in, ch_out, kernel_size, stride
        self.g, self.padding, self.bias = g, padding, bias

    def forward(self, x, fanin=None, fans=None):
        """fanin: kernels.GradFanIn of x (this unit is the consumer of x that runs its backward last, or one of a chain);
        fans: one per part of a list input (this unit runs its backward first and parks the parts' gradients)."""
        if hasattr(self, "conv_bn_fused"):                 # deployed form: BN folded into the conv
            if self._act_name is not None or isinstance(self.act, nn.Identity):
                return kernels.conv_bias_act(x, self.conv_bn_fused, self._act_name)
            return self.act(kernels.conv_bias_act(x, self.conv_bn_fused, None))
        if self._act_name is not None or isinstance(self.act, nn.Identity):
            return kernels.conv_bn_act(x, self.conv, self.norm, self._act_name, None, fanin=fanin, fans=fans)
        return self.act(kernels.conv_bn_act(x, self.conv, self.norm, None, None, fanin=fanin, fans=fans))

    def get_equivalent_kernel_bias(self):
        return _fold_bn(self.conv, self.norm)

    def convert_to_deploy(self):
        if not hasattr(self, "conv"):
            return                                          # already deployed (the reference raises here)
        if not hasattr(self, "conv_bn_fused"):
            self.conv_bn_fused = nn.Conv2d(self.ch_in, self.ch_out, self.kernel_size, self.stride,
                                           groups=self.g, padding=self.padding, bias=True)
        k, b = self.get_equivalent_kernel_bias()
        self.conv_bn_fused.weight.data, self.conv_bn_fused.bias.data = k, b
        del self.conv, self.norm


class ConvNormLayer(nn.Module):
    def __init__(self, ch_in, ch_out, kernel_size, stride, g=1, padding=None, bias=False, act=None):
        super().__init__()
        padding = (kernel_size - 1) // 2 if padding is None else padding
        self.conv = nn.Conv2d(ch_in, ch_out, kernel_size, stride, groups=g, padding=padding, bias=bias)
        self.norm = nn.BatchNorm2d(ch_out)
        self.act = nn.Identity() if act is None else get_activation(act)
        self._act_name = act if isinstance(act, str) else None

    def forward(self, x):
        if self._act_name is not None or isinstance(self.act, nn.Identity):
            return kernels.conv_bn_act(x, self.conv, self.norm, self._act_name, None)
        return self.act(kernels.conv_bn_act(x, self.conv, self.norm, None, None))


class SCDown(nn.Module):
    """1x1 conv+BN then depthwise kxk/s conv+BN (ref hybrid_encoder.py:96-103)."""

    def __init__(self, c1, c2, k, s):
        super().__init__()
        self.cv1 = ConvNormLayer_fuse(c1, c2, 1, 1)
        self.cv2 = ConvNormLayer_fuse(c2, c2, k, s, c2)

    def forward(self, x):
        # (the gradient hand-off of x, if any, is left on the module by the caller: the unit sits inside an nn.Sequential, whose
        # forward - and forward hooks - take no keyword arguments)
        return self.cv2(self.cv1(x, fanin=self.__dict__.pop("_pending_fanin", None)))


class VGGBlock(nn.Module):
    """RepVGG unit: act(conv3x3+BN(x) + conv1x1+BN(x)); re-parameterisable into one 3x3
    (ref hybrid_encoder.py:106-156)."""

    def __init__(self, ch_in, ch_out, act="relu"):
        super().__init__()
        self.ch_in, self.ch_out = ch_in, ch_out
        self.conv1 = ConvNormLayer(ch_in, ch_out, 3, 1, padding=1, act=None)
        self.conv2 = ConvNormLayer(ch_in, ch_out, 1, 1, padding=0, act=None)
        self.act = nn.Identity() if act is None else act

    def forward(self, x, residual=None):
        name = ("silu" if isinstance(self.act, nn.SiLU) else "relu" if isinstance(self.act, nn.ReLU)
                else None if isinstance(self.act, nn.Identity) else False)
        if hasattr(self, "conv"):                          # deployed form: one re-parameterised 3x3
            if name is not False:
                return kernels.conv_bias_act(x, self.conv, name, residual)
            y = self.act(self.conv(x))
            return y if residual is None else y + residual
        if name is not False:
            # both branches, their BatchNorms, the add, the activation (and CSPLayer's residual) as one unit
            return kernels.repvgg_unit(x, self.conv1.conv, self.conv1.norm, self.conv2.conv, self.conv2.norm, name, residual)
        y = self.act(self.conv1(x) + self.conv2(x))
        return y if residual is None else y + residual

    def get_equivalent_kernel_bias(self):
        k3, b3 = _fold_bn(self.conv1.conv, self.conv1.norm)
        k1, b1 = _fold_bn(self.conv2.conv, self.conv2.norm)
        return k3 + F.pad(k1, [1, 1, 1, 1]), b3 + b1

    def convert_to_deploy(self):
        if not hasattr(self, "conv1"):
            return                                          # already deployed
        if not hasattr(self, "conv"):
            self.conv = nn.Conv2d(self.ch_in, self.ch_out, 3, 1, padding=1)
        k, b = self.get_equivalent_kernel_bias()
        self.conv.weight.data, self.conv.bias.data = k, b
        del self.conv1, self.conv2


class CSPLayer(nn.Module):
    """conv3(bottlenecks(conv1(x)) + conv2(x))  (ref hybrid_encoder.py:209-239)."""

    def __init__(self, in_channels, out_channels, num_blocks=3, expansion=1.0, bias=False,
                 act="silu", bottletype=VGGBlock):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = ConvNormLayer_fuse(in_channels, hidden, 1, 1, bias=bias, act=act)
        self.conv2 = ConvNormLayer_fuse(in_channels, hidden, 1, 1, bias=bias, act=act)
        self.bottlenecks = nn.Sequential(
            *[bottletype(hidden, hidden, act=get_activation(act)) for _ in range(num_blocks)])
        self.conv3 = (ConvNormLayer_fuse(hidden, out_channels, 1, 1, bias=bias, act=act)
                      if hidden != out_channels else nn.Identity())

    def forward(self, x, fanin=None):
        """fanin: chain kernels.GradFanIn of x - both 1x1 convolutions add their data gradient onto x's parked gradient."""
        y, r = self.conv1(x, fanin=fanin), self.conv2(x, fanin=fanin)
        blocks = list(self.bottlenecks)
        if blocks and all(isinstance(b, VGGBlock) for b in blocks):
            for b in blocks[:-1]:
                y = b(y)
            return self.conv3(blocks[-1](y, residual=r))       # the residual add rides in the last unit's apply pass
        return self.conv3(self.bottlenecks(y) + r)


class RepNCSPELAN4(nn.Module):
    """cv1 -> split in two -> two chained (CSPLayer + 3x3) branches -> concat all four -> cv4
    (ref hybrid_encoder.py:182-206)."""

    def __init__(self, c1, c2, c3, c4, n=3, bias=False, act="silu"):
        super().__init__()
        self.c = c3 // 2
        self.cv1 = ConvNormLayer_fuse(c1, c3, 1, 1, bias=bias, act=act)
        self.cv2 = nn.Sequential(
            CSPLayer(c3 // 2, c4, n, 1, bias=bias, act=act, bottletype=VGGBlock),
            ConvNormLayer_fuse(c4, c4, 3, 1, bias=bias, act=act))
        self.cv3 = nn.Sequential(
            CSPLayer(c4, c4, n, 1, bias=bias, act=act, bottletype=VGGBlock),
            ConvNormLayer_fuse(c4, c4, 3, 1, bias=bias, act=act))
        self.cv4 = ConvNormLayer_fuse(c3 + (2 * c4), c2, 1, 1, bias=bias, act=act)

    def forward(self, x):
        y1 = self.cv1(x)                                   # x may be a list: the FPN / PAN concat, read in place
        # y1 feeds cv4 whole and cv2's two 1x1 convolutions through its upper half, y2 feeds cv4 and cv3's two: cv4 (created
        # last, backward first) parks their gradients and the branch convolutions add onto them in place (kernels.fan_slice)
        s1, fan1 = kernels.fan_slice(y1, self.c, y1.shape[1] - self.c)
        y2 = self.cv2[1](self.cv2[0](s1, fanin=fan1))
        s2, fan2 = kernels.fan_slice(y2, 0, y2.shape[1])
        y3 = self.cv3[1](self.cv3[0](s2, fanin=fan2))
        return self.cv4([y1, y2, y3], fans=[fan1, fan2, None])     # cat(split(y1), y2, y3) without building it


class MultiheadSelfAttention(nn.Module):
    """Packed-QKV multi-head attention with `nn.MultiheadAttention`'s parameter names
    (in_proj_weight, in_proj_bias, out_proj.{weight,bias}) and batch_first semantics.
    q and k share one input (content + position), v takes the content only; `attn_mask` is
    boolean with True = "may not attend"."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.dropout = dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, qk, value, attn_mask=None):
        return kernels.self_attention(qk, value, self.in_proj_weight, self.in_proj_bias,
                                      self.out_proj.weight, self.out_proj.bias,
                                      self.num_heads, attn_mask)


class TransformerEncoderLayer(nn.Module):
    """Post-LN (or pre-LN) encoder layer (ref hybrid_encoder.py:243-290)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.normalize_before = normalize_before
        self.self_attn = MultiheadSelfAttention(d_model, nhead, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = get_activation(activation)

    @staticmethod
    def with_pos_embed(tensor, pos_embed):
        return tensor if pos_embed is None else tensor + pos_embed

    def forward(self, src, src_mask=None, pos_embed=None) -> torch.Tensor:
        x = self.norm1(src) if self.normalize_before else src
        attn = self.self_attn(self.with_pos_embed(x, pos_embed), x, attn_mask=src_mask)
        if self.normalize_before:
            src = src + self.dropout1(attn)
        else:                                    # post-norm (the reference's configuration): residual + LN in one pass
            src = kernels.add_layer_norm(src, self.dropout1(attn), self.norm1)
        x = self.norm2(src) if self.normalize_before else src
        x = kernels.linear(self.dropout(kernels.linear(x, self.linear1.weight, self.linear1.bias, act=self.activation)),
                           self.linear2.weight, self.linear2.bias)
        if self.normalize_before:
            return src + self.dropout2(x)
        return kernels.add_layer_norm(src, self.dropout2(x), self.norm2)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(copy.deepcopy(encoder_layer) for _ in range(num_layers))
        self.num_layers, self.norm = num_layers, norm

    def forward(self, src, src_mask=None, pos_embed=None) -> torch.Tensor:
        for layer in self.layers:
            src = layer(src, src_mask=src_mask, pos_embed=pos_embed)
        return src if self.norm is None else self.norm(src)


class HybridEncoder(nn.Module):
    __share__ = ["eval_spatial_size"]

    def __init__(self, in_channels=[512, 1024, 2048], feat_strides=[8, 16, 32], hidden_dim=256,
                 nhead=8, dim_feedforward=1024, dropout=0.0, enc_act="gelu", use_encoder_idx=[2],
                 num_encoder_layers=1, pe_temperature=10000, expansion=1.0, depth_mult=1.0,
                 act="silu", eval_spatial_size=None):
        super().__init__()
        self.in_channels, self.feat_strides, self.hidden_dim = in_channels, feat_strides, hidden_dim
        self.use_encoder_idx, self.num_encoder_layers = use_encoder_idx, num_encoder_layers
        self.pe_temperature, self.eval_spatial_size = pe_temperature, eval_spatial_size
        self.out_channels = [hidden_dim] * len(in_channels)
        self.out_strides = feat_strides
        nlev = len(in_channels)

        self.input_proj = nn.ModuleList(
            nn.Sequential(OrderedDict([
                ("conv", nn.Conv2d(c, hidden_dim, kernel_size=1, bias=False)),
                ("norm", nn.BatchNorm2d(hidden_dim))]))
            for c in in_channels)

        layer = TransformerEncoderLayer(hidden_dim, nhead=nhead, dim_feedforward=dim_feedforward,
                                        dropout=dropout, activation=enc_act)
        self.encoder = nn.ModuleList(
            TransformerEncoder(copy.deepcopy(layer), num_encoder_layers)
            for _ in range(len(use_encoder_idx)))

        # NB: `expansion * hidden_dim // 2` is (expansion*hidden_dim)//2 in the reference
        # (hybrid_encoder.py:382,404) - e.g. 21 channels for size n - keep it.
        c4, n_rep = round(expansion * hidden_dim // 2), round(3 * depth_mult)

        def fusion_block():
            return RepNCSPELAN4(hidden_dim * 2, hidden_dim, hidden_dim * 2, c4, n_rep)

        self.lateral_convs = nn.ModuleList()
        self.fpn_blocks = nn.ModuleList()
        for _ in range(nlev - 1):
            self.lateral_convs.append(ConvNormLayer_fuse(hidden_dim, hidden_dim, 1, 1))
            self.fpn_blocks.append(fusion_block())
        self.downsample_convs = nn.ModuleList()
        self.pan_blocks = nn.ModuleList()
        for _ in range(nlev - 1):
            self.downsample_convs.append(nn.Sequential(SCDown(hidden_dim, hidden_dim, 3, 2)))
            self.pan_blocks.append(fusion_block())
        self._pos_cache = {}
        self._reset_parameters()

    def _reset_parameters(self):
        if self.eval_spatial_size:
            for idx in self.use_encoder_idx:
                s = self.feat_strides[idx]
                pe = self.build_2d_sincos_position_embedding(
                    self.eval_spatial_size[1] // s, self.eval_spatial_size[0] // s,
                    self.hidden_dim, self.pe_temperature)
                setattr(self, f"pos_embed{idx}", pe)

    @staticmethod
    def build_2d_sincos_position_embedding(w, h, embed_dim=256, temperature=10000.0):
        """[1, w*h, embed_dim] = [sin(x w), cos(x w), sin(y w), cos(y w)] with the token order
        of meshgrid(arange(w), arange(h), 'ij')  (ref hybrid_encoder.py:425-441)."""
        assert embed_dim % 4 == 0, \
            "Embed dimension must be divisible by 4 for 2D sin-cos position embedding"
        gw, gh = torch.meshgrid(torch.arange(int(w), dtype=torch.float32),
                                torch.arange(int(h), dtype=torch.float32), indexing="ij")
        d = embed_dim // 4
        omega = 1.0 / (temperature ** (torch.arange(d, dtype=torch.float32) / d))
        ow = gw.flatten()[:, None] @ omega[None]
        oh = gh.flatten()[:, None] @ omega[None]
        return torch.concat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)[None]

    def _pos_embed(self, level, w, h, device):
        """Position table for a (w, h) grid, built once per shape and kept on `device`
        (the reference rebuilds it on the host and uploads it every training step,
        hybrid_encoder.py:453-456)."""
        if not (self.training or self.eval_spatial_size is None):
            pe = getattr(self, f"pos_embed{level}")
            if pe.device != device:
                pe = pe.to(device)
                setattr(self, f"pos_embed{level}", pe)
            return pe
        key = (int(w), int(h), str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = self.build_2d_sincos_position_embedding(
                w, h, self.hidden_dim, self.pe_temperature).to(device)
        return self._pos_cache[key]

    def forward(self, feats):
        assert len(feats) == len(self.in_channels)
        proj = [kernels.conv_bn_act(f, p.conv, p.norm, None, None)
                for p, f in zip(self.input_proj, feats)]

        if self.num_encoder_layers > 0:
            for i, lvl in enumerate(self.use_encoder_idx):
                b, c, h, w = proj[lvl].shape
                tokens = proj[lvl].flatten(2).permute(0, 2, 1)
                pe = self._pos_embed(lvl, w, h, tokens.device).to(tokens.dtype)
                mem = self.encoder[i](tokens, pos_embed=pe)
                proj[lvl] = mem.permute(0, 2, 1).reshape(b, c, h, w).contiguous()

        nlev = len(self.in_channels)
        inner = [proj[-1]]
        for idx in range(nlev - 1, 0, -1):
            k = nlev - 1 - idx
            top = self.lateral_convs[k](inner[0])
            inner[0] = top
            # autocast would run the nearest-neighbour copy in fp32 (4x the bytes of the bf16 map it duplicates, plus a cast
            # back for the fusion conv); a pure data movement has nothing to gain from fp32
            with torch.autocast(top.device.type, enabled=False):
                up = kernels.upsample2_nearest(top)
            inner.insert(0, self.fpn_blocks[k]([up, proj[idx - 1]]))

        # outs[0], outs[1] have two consumers: the down-sampling unit of the next PAN level (a 1x1 convolution first) and the
        # decoder (created later = its backward runs first): the decoder's gradient is parked and the 1x1 convolution's data
        # gradient adds onto it in its epilogue instead of autograd adding two maps (kernels.GradFanIn / park_grad)
        outs, fans = [inner[0]], []
        for idx in range(nlev - 1):
            fan = kernels.GradFanIn() if kernels.grad_fanin_enabled(outs[-1]) else None
            self.downsample_convs[idx][0].__dict__["_pending_fanin"] = fan
            down = self.downsample_convs[idx](outs[-1])
            fans.append(fan)
            outs.append(self.pan_blocks[idx]([down, inner[idx + 1]]))
        return [kernels.park_grad(o, f) for o, f in zip(outs, fans + [None])]
