"""Per-size D-FINE hyper-parameters (n/s/m/l/x).

Same nested-dict contract as the reference (`src/d_fine/configs.py:1-213`): `models[size]`
has the keys HGNetv2 / HybridEncoder / DFINETransformer / DFINECriterion / matcher and the
values are passed as **kwargs to the constructors.  Built here from compact per-size rows
instead of five hand-expanded dicts.
"""
from copy import deepcopy

# -- shared defaults (reference configs.py:1-52) -------------------------------------------
base_cfg = {
    "HGNetv2": {"pretrained": False, "local_model_dir": "weight/hgnetv2/", "freeze_stem_only": True},
    "HybridEncoder": {
        "num_encoder_layers": 1, "nhead": 8, "dropout": 0.0, "enc_act": "gelu", "act": "silu",
    },
    "DFINETransformer": {
        "eval_idx": -1, "num_queries": 300, "num_denoising": 100, "label_noise_ratio": 0.5,
        "box_noise_scale": 1.0, "reg_max": 32, "layer_scale": 1,
        "cross_attn_method": "default", "query_select_method": "default",
    },
    "DFINECriterion": {
        "weight_dict": {
            "loss_vfl": 1, "loss_bbox": 5, "loss_giou": 2, "loss_fgl": 0.15, "loss_ddf": 1.5,
            "loss_mask_bce": 1, "loss_mask_dice": 1,
        },
        "losses": ["vfl", "boxes", "local"],
        "alpha": 0.75, "gamma": 2.0, "reg_max": 32,
    },
    "matcher": {
        "weight_dict": {
            "cost_class": 2, "cost_bbox": 5, "cost_giou": 2, "cost_mask": 1, "cost_mask_dice": 1,
        },
        "alpha": 0.25, "gamma": 2.0, "use_focal_loss": True,
    },
}

# size: (backbone, return_idx, freeze_at, freeze_norm, use_lab,
#        enc_in, strides, enc_hidden, enc_idx, enc_ffn, expansion, depth_mult,
#        dec_feat, dec_hidden, dec_layers, reg_scale, points, dec_ffn)
_ROWS = {
    "n": ("B0", [2, 3], -1, False, True,
          [512, 1024], [16, 32], 128, [1], 512, 0.34, 0.5,
          [128, 128], 128, 3, 4, [6, 6], 512),
    "s": ("B0", [1, 2, 3], -1, False, True,
          [256, 512, 1024], [8, 16, 32], 256, [2], 1024, 0.5, 0.34,
          [256, 256, 256], 256, 3, 4, [3, 6, 3], None),
    "m": ("B2", [1, 2, 3], -1, False, True,
          [384, 768, 1536], [8, 16, 32], 256, [2], 1024, 1.0, 0.67,
          [256, 256, 256], 256, 4, 4, [3, 6, 3], 1024),
    "l": ("B4", [1, 2, 3], 0, True, False,
          [512, 1024, 2048], [8, 16, 32], 256, [2], 1024, 1.0, 1.0,
          [256, 256, 256], 256, 6, 4, [3, 6, 3], 1024),
    "x": ("B5", [1, 2, 3], 0, True, False,
          [512, 1024, 2048], [8, 16, 32], 384, [2], 2048, 1.0, 1.0,
          [384, 384, 384], 256, 6, 8, [3, 6, 3], 1024),
}


def _size_cfg(row):
    (bb, ret, fat, fnorm, lab, ein, strides, ehid, eidx, effn, exp, dm,
     dfeat, dhid, dl, rs, pts, dffn) = row
    dec = {
        "feat_channels": list(dfeat), "feat_strides": list(strides), "hidden_dim": dhid,
        "num_levels": len(dfeat), "num_layers": dl, "reg_scale": rs, "num_points": list(pts),
        "mask_dim": 256,
    }
    if dffn is not None:
        dec["dim_feedforward"] = dffn
    return {
        "HGNetv2": {"name": bb, "return_idx": list(ret), "freeze_at": fat,
                    "freeze_norm": fnorm, "use_lab": lab},
        "HybridEncoder": {"in_channels": list(ein), "feat_strides": list(strides),
                          "hidden_dim": ehid, "use_encoder_idx": list(eidx),
                          "dim_feedforward": effn, "expansion": exp, "depth_mult": dm},
        "DFINETransformer": dec,
    }


sizes_cfg = {k: _size_cfg(v) for k, v in _ROWS.items()}
sizes_cfg["m"]["DFINETransformer"]["enable_mask_head"] = False


def merge_configs(base, size_specific):
    """Recursive dict merge, `size_specific` wins (reference configs.py:203-210)."""
    out = dict(base)
    for k, v in size_specific.items():
        out[k] = merge_configs(out[k], v) if isinstance(out.get(k), dict) else v
    return out


# Unlike the reference, every size gets its own deep copy, so `build_loss(..., enable_mask_head
# =True)` cannot leak an appended "masks" loss into other sizes / later calls.
models = {size: merge_configs(deepcopy(base_cfg), cfg) for size, cfg in sizes_cfg.items()}
