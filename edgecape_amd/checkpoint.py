"""Checkpoint handling of the reference.

* `load_checkpoint(model, filename)`       mmcv.runner.load_checkpoint stand-in (test.py:124): torch-pickled
  {'state_dict': ..., 'meta': ...}; key handling in `engine.normalize_state_dict` (encoder_sample.* == encoder_query.*,
  EdgeCape.py:36; fused decoder in_proj -> q/k/v_proj, bias_attn.py:236-265).
* `import_dinov2_hub_state_dict(sd)`       facebookresearch/dinov2 hub checkpoint keys (cls_token, pos_embed, patch_embed.*,
  blocks.N.*, norm.*) -> the `encoder_query.` prefix the detector's state dict uses (EdgeCape.py:35-36).
* `export_pack / load_pack`                flat safetensors pack of the normalised state dict (what `ec_load_tensor` consumes:
  reference key names, fp32), so a deployment does not need torch pickles.
"""
import numpy as np
import torch

from .engine import normalize_state_dict

_HUB_TOP = ("cls_token", "pos_embed", "mask_token", "register_tokens", "patch_embed.", "blocks.", "norm.")


def import_dinov2_hub_state_dict(sd, prefix="encoder_query."):
    out = {}
    for k, v in sd.items():
        if k.startswith(_HUB_TOP):
            out[prefix + k] = v
        else:
            raise KeyError(f"unexpected key in a dinov2 hub state dict: {k}")
    if prefix + "pos_embed" not in out:
        raise KeyError("dinov2 state dict has no pos_embed")
    return out


def merge_state_dicts(backbone_sd, head_ckpt):
    """Released EdgeCape checkpoints already contain encoder_query.*; if a head-only checkpoint is combined with a hub
    backbone, the hub weights fill in whatever encoder_query.* keys are missing."""
    sd = normalize_state_dict(head_ckpt)
    for k, v in import_dinov2_hub_state_dict(backbone_sd).items():
        sd.setdefault(k, v)
    return sd


def load_checkpoint(model, filename, map_location="cpu", strict=True):
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt, strict=strict)
    return ckpt


def _np(v):
    return v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)


def export_pack(state_dict, filename):
    """Normalise keys and write every hot-path tensor (encoder_query.*, keypoint_head_module.*) as fp32 safetensors."""
    from safetensors.numpy import save_file
    sd = normalize_state_dict(state_dict)
    flat = {k: np.ascontiguousarray(_np(v)) for k, v in sd.items()
            if k.startswith("encoder_query.") or k.startswith("keypoint_head_module.")}
    save_file(flat, filename)
    return sorted(flat)


def load_pack(filename):
    from safetensors.numpy import load_file
    return load_file(filename)
