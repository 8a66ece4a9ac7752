"""DFINETransformer: query selection, contrastive denoising, FDR decoder, optional mask head.

Parameter / buffer names and the output-dict contract follow the reference
(`src/d_fine/arch/dfine_decoder.py`).  Differences that matter on MI355X:
  * the encoder memory stays in its native [B, sum(HW), heads, head_dim] layout for the
    deformable gather (no permute/split per step) and the softmax over sampling points plus the
    sampling-location arithmetic are fused into the HIP gather kernel;
  * anchors / valid masks are cached per (shape, device) instead of rebuilt every step.
"""
import copy
import math
from collections import OrderedDict
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from ... import kernels
from .hybrid_encoder import MultiheadSelfAttention
from .utils import (bias_init_with_prob, distance2bbox, get_activation,
                    get_contrastive_denoising_training_group, inverse_sigmoid,
                    weighting_function)

__all__ = ["DFINETransformer"]


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, act="relu"):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.act = get_activation(act)

    def forward(self, x):
        if isinstance(self.act, nn.ReLU) and self.num_layers >= 2:
            return kernels.mlp_relu(x, list(self.layers))       # one autograd node, ReLU backward in the data-gradient epilogues
        for layer in self.layers[:-1]:
            x = kernels.linear(x, layer.weight, layer.bias, act=self.act)
        last = self.layers[-1]
        return kernels.linear(x, last.weight, last.bias)


class MSDeformableAttention(nn.Module):
    """Multi-scale deformable attention without value/output projections
    (ref dfine_decoder.py:49-178)."""

    def __init__(self, embed_dim=256, num_heads=8, num_levels=4, num_points=4, method="default",
                 offset_scale=0.5):
        super().__init__()
        self.embed_dim, self.num_heads, self.num_levels = embed_dim, num_heads, num_levels
        self.offset_scale, self.method = offset_scale, method
        if isinstance(num_points, list):
            assert len(num_points) == num_levels, ""
            self.num_points_list = num_points
        else:
            self.num_points_list = [num_points] * num_levels
        scale = [1 / n for n in self.num_points_list for _ in range(n)]
        self.register_buffer("num_points_scale", torch.tensor(scale, dtype=torch.float32))
        self.total_points = num_heads * sum(self.num_points_list)
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.sampling_offsets = nn.Linear(embed_dim, self.total_points * 2)
        self.attention_weights = nn.Linear(embed_dim, self.total_points)
        self._reset_parameters()
        if method == "discrete":
            for p in self.sampling_offsets.parameters():
                p.requires_grad = False

    def _reset_parameters(self):
        # offsets start as rays: head h points along angle 2*pi*h/H, point p at distance p+1
        init.constant_(self.sampling_offsets.weight, 0)
        ang = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        ray = torch.stack([ang.cos(), ang.sin()], -1)
        ray = ray / ray.abs().max(-1, keepdim=True).values
        ray = ray.reshape(self.num_heads, 1, 2).tile([1, sum(self.num_points_list), 1])
        dist = torch.concat([torch.arange(1, n + 1) for n in self.num_points_list]).reshape(1, -1, 1)
        self.sampling_offsets.bias.data[...] = (ray * dist).flatten()
        init.constant_(self.attention_weights.weight, 0)
        init.constant_(self.attention_weights.bias, 0)

    def forward(self, query, reference_points, value, value_spatial_shapes: List[List[int]]):
        """query [B,Lq,C]; reference_points [B,Lq,1,4] (cxcywh) or [B,Lq,levels,2];
        value [B, sum(HW), heads, head_dim] (native) or the reference's per-level list."""
        bs, lq = query.shape[:2]
        npts = sum(self.num_points_list)
        offsets = kernels.linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).reshape(
            bs, lq, self.num_heads, npts, 2)
        logits = kernels.linear(query, self.attention_weights.weight, self.attention_weights.bias).reshape(
            bs, lq, self.num_heads, npts)
        if isinstance(value, (list, tuple)):
            value = torch.cat(list(value), dim=-1).permute(0, 3, 1, 2).contiguous()

        if reference_points.shape[-1] == 4:
            # loc = ref_xy + offset * (1/n_points_of_level) * ref_wh * offset_scale ;
            # weights = softmax over all points - both fused into the gather kernel.
            return kernels.msda_fused(value, value_spatial_shapes, reference_points.reshape(bs, lq, 4),
                                      offsets, logits, self.num_points_list, self.offset_scale)
        if reference_points.shape[-1] == 2:
            norm = torch.tensor(value_spatial_shapes, device=query.device).flip([1])
            norm = norm.repeat_interleave(torch.tensor(self.num_points_list, device=query.device), 0)
            ref = reference_points.repeat_interleave(
                torch.tensor(self.num_points_list, device=query.device), 2)
            loc = ref[:, :, None] + offsets / norm.reshape(1, 1, 1, npts, 2)
            return kernels.msda(value, value_spatial_shapes, loc, F.softmax(logits, -1),
                                self.num_points_list)
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))


class Gate(nn.Module):
    """LN(sigmoid(W[x1;x2])_1 * x1 + sigmoid(W[x1;x2])_2 * x2)  (ref dfine_decoder.py:258-271)."""

    def __init__(self, d_model):
        super().__init__()
        self.gate = nn.Linear(2 * d_model, 2 * d_model)
        init.constant_(self.gate.bias, bias_init_with_prob(0.5))
        init.constant_(self.gate.weight, 0)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, x1, x2):
        # x1 may come as two aliases of one tensor (kernels.fan_out: the gradients of all its consumers are summed in one pass)
        xa, xb = x1 if isinstance(x1, tuple) else (x1, x1)
        g = kernels.linear(torch.cat([xa, x2], dim=-1), self.gate.weight, self.gate.bias)
        return kernels.gate_layer_norm(g, xb, x2, self.norm)


class TransformerDecoderLayer(nn.Module):
    """self-attn -> LN -> deformable cross-attn -> Gate -> FFN -> clamp -> LN
    (ref dfine_decoder.py:181-255)."""

    def __init__(self, d_model=256, n_head=8, dim_feedforward=1024, dropout=0.0, activation="relu",
                 n_levels=4, n_points=4, cross_attn_method="default", layer_scale=None):
        super().__init__()
        if layer_scale is not None:
            dim_feedforward = round(layer_scale * dim_feedforward)
            d_model = round(layer_scale * d_model)
        self.self_attn = MultiheadSelfAttention(d_model, n_head, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformableAttention(d_model, n_head, n_levels, n_points,
                                                method=cross_attn_method)
        self.dropout2 = nn.Dropout(dropout)
        self.gateway = Gate(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.activation = get_activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        init.xavier_uniform_(self.linear1.weight)
        init.xavier_uniform_(self.linear2.weight)

    @staticmethod
    def with_pos_embed(tensor, pos):
        if pos is None:
            return tensor
        if pos.dtype != tensor.dtype and not torch.is_grad_enabled():
            # inference: ATen's mixed-type add (fp32 stream + bf16 embedding) takes ~50 us on a [1, 300, 256] tensor against
            # 3.5 + 2.7 us for a cast and a same-type add (tools/probe/infer_profile.py: 8 x per forward, 0.4 of 3.0 ms at batch 1)
            pos = pos.to(tensor.dtype)
        return tensor + pos

    def forward_ffn(self, tgt):
        if isinstance(self.activation, nn.ReLU) and (self.dropout3.p == 0.0 or not self.training):
            return kernels.mlp_relu(tgt, [self.linear1, self.linear2])
        h = self.dropout3(kernels.linear(tgt, self.linear1.weight, self.linear1.bias, act=self.activation))
        return kernels.linear(h, self.linear2.weight, self.linear2.bias)

    def forward(self, target, reference_points, value, spatial_shapes, attn_mask=None,
                query_pos_embed=None):
        # the fp32 token stream has three consumers here and three behind the first LayerNorm: one alias each, so that their
        # gradients are summed in ONE pass instead of pairwise by autograd (kernels.fan_out)
        t_qk, t_v, t_res = kernels.fan_out(target, 3)
        qk = self.with_pos_embed(t_qk, query_pos_embed)
        target = kernels.add_layer_norm(t_res, self.dropout1(self.self_attn(qk, t_v, attn_mask=attn_mask)),
                                        self.norm1)
        t_q, t_cat, t_gate = kernels.fan_out(target, 3)
        cross = self.cross_attn(self.with_pos_embed(t_q, query_pos_embed), reference_points,
                                value, spatial_shapes)
        target = self.gateway((t_cat, t_gate), self.dropout2(cross))
        return kernels.add_layer_norm(target, self.dropout4(self.forward_ffn(target)), self.norm3, clamp=65504.0)


class Integral(nn.Module):
    """sum_n softmax(logits)_n * W(n) per box edge (ref dfine_decoder.py:274-295)."""

    def __init__(self, reg_max=32):
        super().__init__()
        self.reg_max = reg_max

    def forward(self, x, project):
        lead = list(x.shape[:-1])
        # always fp32: under autocast the reference runs this 33-term dot product in half
        # precision (F.linear is autocast-eligible), which costs ~3 digits of box position
        with torch.autocast(x.device.type, enabled=False):
            p = F.softmax(x.float().reshape(-1, self.reg_max + 1), dim=1)
            return F.linear(p, project.to(x.device).float()).reshape(lead + [-1])


class LQE(nn.Module):
    """Location-quality estimator: logits += MLP(top-k bin probabilities and their mean)
    (ref dfine_decoder.py:298-313)."""

    def __init__(self, k, hidden_dim, num_layers, reg_max):
        super().__init__()
        self.k, self.reg_max = k, reg_max
        self.reg_conf = MLP(4 * (k + 1), hidden_dim, 1, num_layers)
        init.constant_(self.reg_conf.layers[-1].bias, 0)
        init.constant_(self.reg_conf.layers[-1].weight, 0)

    def forward(self, scores, pred_corners):
        b, l, _ = pred_corners.size()
        prob = F.softmax(pred_corners.reshape(b, l, 4, self.reg_max + 1), dim=-1)
        top, _ = prob.topk(self.k, dim=-1)
        stat = torch.cat([top, top.mean(dim=-1, keepdim=True)], dim=-1)
        return scores + self.reg_conf(stat.reshape(b, l, -1))


class MaskDecoder(nn.Module):
    """Fuses the PAN maps into 1/4-resolution mask features (ref dfine_decoder.py:316-370)."""

    def __init__(self, in_chs, out_ch=256):
        super().__init__()
        groups = 32
        self.lateral = nn.ModuleList(nn.Conv2d(c, out_ch, 1, bias=False) for c in in_chs)
        self.bn = nn.ModuleList(nn.GroupNorm(groups, out_ch) for _ in in_chs)
        self.fusion_conv = nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False)
        self.fusion_norm = nn.GroupNorm(groups, out_ch)
        self.up_conv = nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False)
        self.bn1 = nn.GroupNorm(groups, out_ch)
        self.act = nn.ReLU(inplace=True)
        init.kaiming_normal_(self.up_conv.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, feats):
        # every piece is a HIP operator on CUDA tensors (MFMA convolutions under bf16 autocast, GroupNorm [+ ReLU] and the
        # bilinear upsample [+ sum] as one pass each: csrc/mask.hip) and the reference's ATen composition otherwise
        x = kernels.group_norm_act(kernels.conv_plain(feats[0], self.lateral[0]), self.bn[0])
        size = x.shape[-2:]
        for lat, gn, f in zip(self.lateral[1:], self.bn[1:], feats[1:]):
            x = kernels.bilinear_resize(kernels.group_norm_act(kernels.conv_plain(f, lat), gn), size, base=x)
        x = kernels.group_norm_act(kernels.conv_plain(x, self.fusion_conv), self.fusion_norm, relu=True)
        x = kernels.bilinear_resize(x, (2 * x.shape[-2], 2 * x.shape[-1]))
        return kernels.group_norm_act(kernels.conv_plain(x, self.up_conv), self.bn1, relu=True)


class TransformerDecoder(nn.Module):
    """Stack of decoder layers with fine-grained distribution refinement
    (ref dfine_decoder.py:373-524)."""

    def __init__(self, hidden_dim, decoder_layer, decoder_layer_wide, num_layers, num_head, reg_max,
                 reg_scale, up, eval_idx=-1, layer_scale=2):
        super().__init__()
        self.hidden_dim, self.num_layers, self.layer_scale = hidden_dim, num_layers, layer_scale
        self.num_head = num_head
        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx
        self.up, self.reg_scale, self.reg_max = up, reg_scale, reg_max
        self.layers = nn.ModuleList(
            [copy.deepcopy(decoder_layer) for _ in range(self.eval_idx + 1)]
            + [copy.deepcopy(decoder_layer_wide) for _ in range(num_layers - self.eval_idx - 1)])
        self.lqe_layers = nn.ModuleList(copy.deepcopy(LQE(4, 64, 2, reg_max))
                                        for _ in range(num_layers))

    def value_op(self, memory, value_proj, value_scale, memory_mask, memory_spatial_shapes):
        """Memory [B, L, C] -> [B, L, heads, head_dim] view for the gather kernel.  (The
        reference permutes to per-level [B, heads, head_dim, HW] lists for grid_sample,
        dfine_decoder.py:410-420; its value_proj / value_scale arguments are never used at
        layer_scale=1.)"""
        value = value_proj(memory) if value_proj is not None else memory
        if value_scale is not None:
            value = F.interpolate(memory, size=value_scale)
        if memory_mask is not None:
            value = value * memory_mask.to(value.dtype).unsqueeze(-1)
        return value.reshape(value.shape[0], value.shape[1], self.num_head, -1)

    def _fdr_constants(self, project, reg_scale):
        """W(n) and reg_scale as host numbers for the HIP FDR kernel (model constants: one D2H)."""
        key = (project.data_ptr() if hasattr(self, "project") else (self.up._version, self.reg_scale._version,
                                                                    self.up.data_ptr()), float(self.reg_max))
        cache = getattr(self, "_fdr_cache", None)
        if cache is None or cache[0] != key:
            self._fdr_cache = (key, project.detach().float().cpu().tolist(),
                               float(reg_scale.detach().float().cpu()) if torch.is_tensor(reg_scale) else float(reg_scale))
        return self._fdr_cache[1], self._fdr_cache[2]

    def convert_to_deploy(self):
        self.project = weighting_function(self.reg_max, self.up, self.reg_scale, deploy=True)
        self.layers = self.layers[: self.eval_idx + 1]
        self.lqe_layers = nn.ModuleList(
            [nn.Identity()] * self.eval_idx + [self.lqe_layers[self.eval_idx]])

    def forward(self, target, ref_points_unact, memory, spatial_shapes, bbox_head, score_head,
                query_pos_head, pre_bbox_head, integral, up, reg_scale, attn_mask=None,
                memory_mask=None, return_queries: bool = False):
        value = self.value_op(memory, None, None, memory_mask, spatial_shapes)
        if self.training and self.layer_scale == 1:
            value = kernels.msda_share_value_grad(value)      # every layer gathers from this one tensor
        project = self.project if hasattr(self, "project") else weighting_function(
            self.reg_max, up, reg_scale)

        boxes, logits, corners, refs = [], [], [], []
        queries = [] if return_queries else None
        ref_detach = F.sigmoid(ref_points_unact)
        out = target
        out_detach = prev_corners = 0

        for i, layer in enumerate(self.layers):
            pos = kernels.clamp_pos(query_pos_head(ref_detach))          # .clamp(min=-10, max=10)
            if i >= self.eval_idx + 1 and self.layer_scale > 1:  # "wide" layers (dead at scale 1)
                pos = F.interpolate(pos, scale_factor=self.layer_scale)
                value = self.value_op(memory, None, pos.shape[-1], memory_mask, spatial_shapes)
                out = F.interpolate(out, size=pos.shape[-1])
                out_detach = out.detach()

            out = layer(out, ref_detach.unsqueeze(2), value, spatial_shapes, attn_mask, pos)
            if return_queries:
                queries.append(out)
            # the layer output feeds the heads below and the next layer: one alias per consumer (kernels.fan_out; aliases that
            # stay unused bring no gradient)
            taps = iter(kernels.fan_out(out, 5)) if (self.training and torch.is_grad_enabled()) else None
            base = out

            def tap():
                return next(taps) if taps is not None else base

            if i == 0:
                # classic sigmoid-space box head on the first layer seeds the FDR reference
                # (.float(): ATen's mixed bf16 + fp32 add takes 45 - 90 us on a [B, Q, 4] tensor here, the cast + fp32 add 10 us -
                # tools/probe/tiny_add.py; the promoted sum is the same number)
                pre_bboxes = F.sigmoid(pre_bbox_head(tap()).float() + inverse_sigmoid(ref_detach))
                # (after deploy() the heads before eval_idx are nn.Identity placeholders: ref dfine_decoder.py:698-707)
                pre_scores = kernels.linear(tap(), score_head[0].weight, score_head[0].bias) \
                    if isinstance(score_head[0], nn.Linear) else score_head[0](tap())
                ref_initial = pre_bboxes.detach()

            # FDR: residual update of the edge distributions, decoded around the initial box
            pred_corners = bbox_head[i](tap() + out_detach) + prev_corners
            # Integral + distance2bbox (+ the LQE statistics) in one HIP kernel; after deploy() the layers in front of eval_idx
            # carry an nn.Identity in place of their LQE (ref dfine_decoder.py:422-427) and only take the boxes
            fused_box = pred_corners.is_cuda and self.reg_max == 32
            fused = fused_box and isinstance(self.lqe_layers[i], LQE) and self.lqe_layers[i].k == 4
            if fused_box:
                wtable, rs = self._fdr_constants(project, reg_scale)
                box, stat = kernels.fdr_decode(pred_corners, ref_initial, wtable, rs)
            else:
                box = distance2bbox(ref_initial, integral(pred_corners, project), reg_scale)

            if self.training or i == self.eval_idx:
                if fused:
                    scores = kernels.linear(tap(), score_head[i].weight, score_head[i].bias) \
                        + self.lqe_layers[i].reg_conf(stat)
                else:
                    scores = self.lqe_layers[i](score_head[i](tap()), pred_corners)
                logits.append(scores)
                boxes.append(box)
                corners.append(pred_corners)
                refs.append(ref_initial)
                if not self.training:
                    break

            prev_corners = pred_corners
            ref_detach = box.detach()
            out_detach = out.detach()
            out = tap()                                  # the next layer's input

        hs = torch.stack(queries) if return_queries else None
        # per-layer LISTS, not stacked tensors (the reference stacks, dfine_decoder.py:522-524, then indexes the layers apart
        # again): a select on a stacked tensor costs a zero fill + copy + add per layer and head tensor in backward
        if kernels.STACK_LAYER_OUTPUTS:          # (the reference's form; tools/ab_step.py kernels.STACK_LAYER_OUTPUTS for the A/B)
            return torch.stack(boxes), torch.stack(logits), torch.stack(corners), torch.stack(refs), pre_bboxes, pre_scores, hs
        return boxes, logits, corners, refs, pre_bboxes, pre_scores, hs


class DFINETransformer(nn.Module):
    __share__ = ["num_classes", "eval_spatial_size"]

    def __init__(self, num_classes=80, hidden_dim=256, num_queries=300,
                 feat_channels=[512, 1024, 2048], feat_strides=[8, 16, 32], num_levels=3,
                 num_points=4, nhead=8, num_layers=6, dim_feedforward=1024, dropout=0.0,
                 activation="relu", num_denoising=100, label_noise_ratio=0.5, box_noise_scale=1.0,
                 learn_query_content=False, eval_spatial_size=None, eval_idx=-1, eps=1e-2,
                 aux_loss=True, cross_attn_method="default", query_select_method="default",
                 reg_max=32, reg_scale=4.0, layer_scale=1, enable_mask_head=False, mask_dim=256):
        super().__init__()
        assert len(feat_channels) <= num_levels
        assert len(feat_strides) == len(feat_channels)
        for _ in range(num_levels - len(feat_strides)):
            feat_strides.append(feat_strides[-1] * 2)
        assert query_select_method in ("default", "one2many", "agnostic"), ""
        assert cross_attn_method in ("default", "discrete"), ""

        self.hidden_dim, self.nhead, self.feat_strides = hidden_dim, nhead, feat_strides
        self.num_levels, self.num_classes, self.num_queries = num_levels, num_classes, num_queries
        self.eps, self.num_layers, self.eval_spatial_size = eps, num_layers, eval_spatial_size
        self.aux_loss, self.reg_max, self.mask_dim = aux_loss, reg_max, mask_dim
        self.enable_mask_head = enable_mask_head
        self.cross_attn_method, self.query_select_method = cross_attn_method, query_select_method
        scaled_dim = round(layer_scale * hidden_dim)

        self._build_input_proj_layer(feat_channels)

        self.up = nn.Parameter(torch.tensor([0.5]), requires_grad=False)
        self.reg_scale = nn.Parameter(torch.tensor([reg_scale]), requires_grad=False)
        layer_args = (hidden_dim, nhead, dim_feedforward, dropout, activation, num_levels, num_points)
        self.decoder = TransformerDecoder(
            hidden_dim,
            TransformerDecoderLayer(*layer_args, cross_attn_method=cross_attn_method),
            TransformerDecoderLayer(*layer_args, cross_attn_method=cross_attn_method,
                                    layer_scale=layer_scale),
            num_layers, nhead, reg_max, self.reg_scale, self.up, eval_idx, layer_scale)

        self.num_denoising = num_denoising
        self.label_noise_ratio, self.box_noise_scale = label_noise_ratio, box_noise_scale
        if num_denoising > 0:
            self.denoising_class_embed = nn.Embedding(num_classes + 1, hidden_dim,
                                                      padding_idx=num_classes)
            init.normal_(self.denoising_class_embed.weight[:-1])

        if enable_mask_head:
            self.mask_decoder = MaskDecoder(in_chs=feat_channels, out_ch=mask_dim)
            self.mask_head = MLP(hidden_dim, hidden_dim, mask_dim, num_layers=3)

        self.learn_query_content = learn_query_content
        if learn_query_content:
            self.tgt_embed = nn.Embedding(num_queries, hidden_dim)
        self.query_pos_head = MLP(4, 2 * hidden_dim, hidden_dim, 2)
        self.enc_output = nn.Sequential(OrderedDict([
            ("proj", nn.Linear(hidden_dim, hidden_dim)), ("norm", nn.LayerNorm(hidden_dim))]))
        self.enc_score_head = nn.Linear(
            hidden_dim, 1 if query_select_method == "agnostic" else num_classes)
        self.enc_bbox_head = MLP(hidden_dim, hidden_dim, 4, 3)

        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx
        n_std, n_wide = self.eval_idx + 1, num_layers - self.eval_idx - 1
        self.dec_score_head = nn.ModuleList(
            [nn.Linear(hidden_dim, num_classes) for _ in range(n_std)]
            + [nn.Linear(scaled_dim, num_classes) for _ in range(n_wide)])
        self.pre_bbox_head = MLP(hidden_dim, hidden_dim, 4, 3)
        nbins = 4 * (reg_max + 1)
        self.dec_bbox_head = nn.ModuleList(
            [MLP(hidden_dim, hidden_dim, nbins, 3) for _ in range(n_std)]
            + [MLP(scaled_dim, scaled_dim, nbins, 3) for _ in range(n_wide)])
        self.integral = Integral(reg_max)

        if eval_spatial_size:
            anchors, valid_mask = self._generate_anchors()
            self.register_buffer("anchors", anchors)
            self.register_buffer("valid_mask", valid_mask)
        self._anchor_cache = {}
        self._reset_parameters(feat_channels)

    # ------------------------------------------------------------------ construction helpers
    def _build_input_proj_layer(self, feat_channels):
        def proj(cin, k, s, p):
            return nn.Sequential(OrderedDict([
                ("conv", nn.Conv2d(cin, self.hidden_dim, k, s, padding=p, bias=False)),
                ("norm", nn.BatchNorm2d(self.hidden_dim))]))

        self.input_proj = nn.ModuleList(
            nn.Identity() if c == self.hidden_dim else proj(c, 1, 1, 0) for c in feat_channels)
        cin = feat_channels[-1]
        for _ in range(self.num_levels - len(feat_channels)):
            if cin == self.hidden_dim:
                self.input_proj.append(nn.Identity())
            else:
                self.input_proj.append(proj(cin, 3, 2, 1))
                cin = self.hidden_dim

    def _reset_parameters(self, feat_channels):
        prior = bias_init_with_prob(0.01)
        init.constant_(self.enc_score_head.bias, prior)
        for head in (self.enc_bbox_head, self.pre_bbox_head):
            init.constant_(head.layers[-1].weight, 0)
            init.constant_(head.layers[-1].bias, 0)
        for cls_, reg_ in zip(self.dec_score_head, self.dec_bbox_head):
            init.constant_(cls_.bias, prior)
            if hasattr(reg_, "layers"):
                init.constant_(reg_.layers[-1].weight, 0)
                init.constant_(reg_.layers[-1].bias, 0)
        init.xavier_uniform_(self.enc_output[0].weight)
        if self.learn_query_content:
            init.xavier_uniform_(self.tgt_embed.weight)
        init.xavier_uniform_(self.query_pos_head.layers[0].weight)
        init.xavier_uniform_(self.query_pos_head.layers[1].weight)
        for m, cin in zip(self.input_proj, feat_channels):
            if cin != self.hidden_dim:
                init.xavier_uniform_(m[0].weight)

    def convert_to_deploy(self):
        self.dec_score_head = nn.ModuleList(
            [nn.Identity()] * self.eval_idx + [self.dec_score_head[self.eval_idx]])
        self.dec_bbox_head = nn.ModuleList(
            [h if i <= self.eval_idx else nn.Identity() for i, h in enumerate(self.dec_bbox_head)])

    # ------------------------------------------------------------------ encoder-side inputs
    def _get_encoder_input(self, feats: List[torch.Tensor]):
        def apply(p, f):
            return f if isinstance(p, nn.Identity) else kernels.conv_bn_act(f, p.conv, p.norm, None, None)

        proj = [apply(p, f) for p, f in zip(self.input_proj, feats)]
        for i in range(len(proj), self.num_levels):
            proj.append(apply(self.input_proj[i], feats[-1] if i == len(feats) else proj[-1]))
        shapes = [[f.shape[2], f.shape[3]] for f in proj]
        memory = kernels.flatten_levels(proj)
        return memory, shapes

    def _generate_anchors(self, spatial_shapes=None, grid_size=0.05, dtype=torch.float32,
                          device="cpu"):
        """Logit-space anchors [1, sum(HW), 4] (cell centre, size grid_size * 2^level) and
        their validity mask (ref dfine_decoder.py:803-826)."""
        if spatial_shapes is None:
            eh, ew = self.eval_spatial_size
            spatial_shapes = [[int(eh / s), int(ew / s)] for s in self.feat_strides]
        per_level = []
        for lvl, (h, w) in enumerate(spatial_shapes):
            gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            xy = (torch.stack([gx, gy], dim=-1).unsqueeze(0) + 0.5) / torch.tensor([w, h], dtype=dtype)
            wh = torch.ones_like(xy) * grid_size * (2.0 ** lvl)
            per_level.append(torch.concat([xy, wh], dim=-1).reshape(-1, h * w, 4))
        anchors = torch.concat(per_level, dim=1).to(device)
        valid = ((anchors > self.eps) * (anchors < 1 - self.eps)).all(-1, keepdim=True)
        anchors = torch.where(valid, torch.log(anchors / (1 - anchors)), torch.inf)
        return anchors, valid

    def _anchors_for(self, spatial_shapes, device):
        if not (self.training or self.eval_spatial_size is None):
            return self.anchors, self.valid_mask
        key = (tuple(map(tuple, spatial_shapes)), str(device))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = self._generate_anchors(spatial_shapes, device=device)
        return self._anchor_cache[key]

    def _invalid_rows(self, spatial_shapes, valid):
        """Indices of the anchors outside (eps, 1 - eps) (ref dfine_decoder.py:803-826): made once per anchor set and kept next to
        it, keyed like `_anchor_cache` (the lookup is a host sync; the registered `valid_mask` buffer of eval mode has its own entry)."""
        cache = self.__dict__.setdefault("_invalid_cache", {})
        key = (tuple(map(tuple, spatial_shapes)), str(valid.device), bool(self.training or self.eval_spatial_size is None))
        if key not in cache:
            cache[key] = (~valid.reshape(-1).bool()).nonzero().reshape(-1)
        return cache[key]

    def _enc_output(self, t):
        """enc_output = Linear + LayerNorm (ref dfine_decoder.py:615-621), both HIP kernels on the GPU."""
        return kernels.layer_norm(kernels.linear_module(self.enc_output[0], t), self.enc_output[1])

    def _enc_scores(self, t):
        return kernels.linear_module(self.enc_score_head, t)

    def _get_decoder_input(self, memory, spatial_shapes, denoising_logits=None,
                           denoising_bbox_unact=None):
        anchors, valid = self._anchors_for(spatial_shapes, memory.device)
        if memory.shape[0] > 1:
            anchors = anchors.expand(memory.shape[0], -1, -1)
        keep = valid.to(memory.dtype)                       # [1, sum(HW), 1]: 0 for anchors outside (0.01, 0.99)
        if self.training and torch.is_grad_enabled():
            # Query selection only needs the scores of all anchors to pick indices; every op on this
            # path (Linear, LayerNorm, score head) is row-wise and gradients only flow through the
            # selected rows.  So: score all sum(HW) rows WITHOUT autograd, then recompute the 300
            # selected rows with autograd.  Same values and gradients as the reference
            # (dfine_decoder.py:842-853), but the backward no longer runs three GEMMs + a LayerNorm
            # over B*8400 mostly-zero gradient rows.  The validity mask is applied to the whole memory for the
            # scoring pass only (no autograd) and to the 300 gathered rows on the differentiable path: the same
            # products, without a full-size multiply (and its saved operand) in the backward.
            with torch.no_grad():
                # every op of the scoring pass is row-wise and a masked row is all zeros: score the memory as it stands and give
                # the few masked rows (the border anchors of the finest level) the score of a zero row afterwards - the same
                # numbers as scoring keep * memory without the full-size multiply (86 us per step for D-FINE-m)
                inv = self._invalid_rows(spatial_shapes, valid)
                scores_all = self._enc_scores(self._enc_output(memory))
                if inv.numel():
                    zero_score = self._enc_scores(self._enc_output(memory.new_zeros(1, 1, memory.shape[-1])))
                    scores_all[:, inv] = zero_score.to(scores_all.dtype)
            ind = self._topk_indices(scores_all, self.num_queries)

            def take(t):
                return t.gather(dim=1, index=ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))

            # one autograd node hands out the selected rows AND the memory for the decoder's value path (forward() picks it up):
            # its backward adds the rows' gradient onto the value path's in place instead of two full-size tensors being added
            value_memory, rows = kernels.take_rows_and_pass(memory, ind)
            top_mem = self._enc_output(rows * take(keep.expand(memory.shape[0], -1, -1)))
            top_logits = self._enc_scores(top_mem)
            top_anchor = take(anchors)
        else:
            memory = keep * memory
            value_memory = None
            out_mem = self._enc_output(memory)
            enc_logits = self._enc_scores(out_mem)
            top_mem, top_logits, top_anchor = self._select_topk(out_mem, enc_logits, anchors,
                                                                self.num_queries)
        box_unact = self.enc_bbox_head(top_mem).float() + top_anchor      # (.float(): see the note in TransformerDecoder.forward)
        enc_boxes, enc_logits_list = [], []
        if self.training:
            enc_boxes.append(F.sigmoid(box_unact))
            enc_logits_list.append(top_logits)

        if self.learn_query_content:
            content = self.tgt_embed.weight.unsqueeze(0).tile([memory.shape[0], 1, 1])
        else:
            content = top_mem.detach()
        box_unact = box_unact.detach()
        if denoising_bbox_unact is not None:
            box_unact = torch.concat([denoising_bbox_unact, box_unact], dim=1)
            content = torch.concat([denoising_logits, content], dim=1)
        # (value_memory: the memory for the decoder's value path, through the query selection's autograd node - None: the caller's)
        return content, box_unact, enc_boxes, enc_logits_list, value_memory

    def _topk_indices(self, outputs_logits, topk: int):
        if self.query_select_method == "default":
            return kernels.topk_anchors(outputs_logits, topk)      # max over classes + top-k, fused
        if self.query_select_method == "one2many":
            score = outputs_logits.flatten(1)
        else:
            score = outputs_logits.squeeze(-1)
        ind = kernels.topk_indices(score, topk)
        if self.query_select_method == "one2many":
            ind = ind // self.num_classes
        return ind

    def _select_topk(self, memory, outputs_logits, outputs_anchors_unact, topk: int):
        ind = self._topk_indices(outputs_logits, topk)

        def take(t):
            return t.gather(dim=1, index=ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))

        return (take(memory), take(outputs_logits) if self.training else None,
                take(outputs_anchors_unact))

    # ------------------------------------------------------------------ masks
    def _should_do_masks(self, targets):
        if not self.enable_mask_head:
            return False
        if targets is None:
            return True
        for t in targets:
            m = t.get("masks", None)
            if m is not None and hasattr(m, "numel") and m.numel() > 0:
                return True
        return False

    def _mask_logits_from_h(self, h, mask_feat):
        emb = self.mask_head(h)
        emb = emb * (emb.shape[-1] ** -0.5)
        return kernels.mask_logits(emb, mask_feat)

    # ------------------------------------------------------------------ forward
    def forward(self, feats, targets=None):
        do_masks = self._should_do_masks(targets)
        memory, spatial_shapes = self._get_encoder_input(feats)

        dn_logits = dn_boxes = attn_mask = dn_meta = None
        if self.training and self.num_denoising > 0:
            dn_logits, dn_boxes, attn_mask, dn_meta = get_contrastive_denoising_training_group(
                targets, self.num_classes, self.num_queries, self.denoising_class_embed,
                num_denoising=self.num_denoising, label_noise_ratio=self.label_noise_ratio,
                box_noise_scale=1.0)  # the reference hard-codes 1.0 here (dfine_decoder.py:948)

        content, ref_unact, enc_boxes, enc_logits, value_memory = self._get_decoder_input(
            memory, spatial_shapes, dn_logits, dn_boxes)
        if value_memory is not None:
            memory = value_memory

        out_bboxes, out_logits, out_corners, out_refs, pre_bboxes, pre_logits, hs = self.decoder(
            content, ref_unact, memory, spatial_shapes, self.dec_bbox_head, self.dec_score_head,
            self.query_pos_head, self.pre_bbox_head, self.integral, self.up, self.reg_scale,
            attn_mask=attn_mask, return_queries=do_masks)

        has_dn = self.training and dn_meta is not None
        dn_hs = None
        if has_dn:
            split = dn_meta["dn_num_split"]
            dn_pre_logits, pre_logits = torch.split(pre_logits, split, dim=1)
            dn_pre_bboxes, pre_bboxes = torch.split(pre_bboxes, split, dim=1)
            def split_layers(layers):             # every layer's [B, dn + Q, .] into its denoising / matching parts (views)
                if torch.is_tensor(layers):
                    return torch.split(layers, split, dim=2)
                parts = [torch.split(t, split, dim=1) for t in layers]
                return [p[0] for p in parts], [p[1] for p in parts]

            dn_out_bboxes, out_bboxes = split_layers(out_bboxes)
            dn_out_logits, out_logits = split_layers(out_logits)
            dn_out_corners, out_corners = split_layers(out_corners)
            dn_out_refs, out_refs = split_layers(out_refs)
            if do_masks and hs is not None:
                dn_hs, hs = torch.split(hs, split, dim=2)

        pred_masks = aux_masks = dn_pred_masks = dn_aux_masks = None
        if do_masks:
            mask_feat = self.mask_decoder(feats)
            pred_masks = self._mask_logits_from_h(hs[-1], mask_feat)
            aux_masks = [self._mask_logits_from_h(h, mask_feat) for h in hs[:-1]]
            if has_dn and dn_hs is not None:
                dn_pred_masks = self._mask_logits_from_h(dn_hs[-1], mask_feat)
                dn_aux_masks = [self._mask_logits_from_h(h, mask_feat) for h in dn_hs[:-1]]

        out = {"pred_logits": out_logits[-1], "pred_boxes": out_bboxes[-1]}
        if self.training:
            out.update(pred_corners=out_corners[-1], ref_points=out_refs[-1], up=self.up,
                       reg_scale=self.reg_scale)
            if do_masks:
                out["pred_masks"] = pred_masks
        elif do_masks:
            out["pred_masks"] = torch.sigmoid(pred_masks)

        if self.training and self.aux_loss:
            out["aux_outputs"] = self._set_aux_loss2(
                out_logits[:-1], out_bboxes[:-1], out_corners[:-1], out_refs[:-1],
                out_corners[-1], out_logits[-1], aux_masks=aux_masks if do_masks else None)
            out["enc_aux_outputs"] = self._set_aux_loss(enc_logits, enc_boxes)
            out["pre_outputs"] = {"pred_logits": pre_logits, "pred_boxes": pre_bboxes}
            out["enc_meta"] = {"class_agnostic": self.query_select_method == "agnostic"}
            if dn_meta is not None:
                out["dn_outputs"] = self._set_aux_loss2(
                    dn_out_logits, dn_out_bboxes, dn_out_corners, dn_out_refs,
                    dn_out_corners[-1], dn_out_logits[-1],
                    aux_masks=dn_aux_masks if do_masks else None)
                if do_masks and dn_pred_masks is not None:
                    out["dn_pred_masks"] = dn_pred_masks
                out["dn_pre_outputs"] = {"pred_logits": dn_pre_logits, "pred_boxes": dn_pre_bboxes}
                out["dn_meta"] = dn_meta
        return out

    @staticmethod
    def _set_aux_loss(outputs_class, outputs_coord):
        return [{"pred_logits": a, "pred_boxes": b} for a, b in zip(outputs_class, outputs_coord)]

    @staticmethod
    def _set_aux_loss2(outputs_class, outputs_coord, outputs_corners, outputs_ref,
                       teacher_corners=None, teacher_logits=None, aux_masks=None):
        rows = zip(outputs_class, outputs_coord, outputs_corners, outputs_ref,
                   aux_masks if aux_masks is not None else [None] * len(outputs_class))
        res = []
        for a, b, c, d, m in rows:
            item = {"pred_logits": a, "pred_boxes": b, "pred_corners": c, "ref_points": d,
                    "teacher_corners": teacher_corners, "teacher_logits": teacher_logits}
            if aux_masks is not None:
                item["pred_masks"] = m
            res.append(item)
        return res
