"""Importer for the reference's real checkpoints (SURVEY.md §5 "Checkpoint / resume", §8f rank 4).

Layouts read (none of these files ships offline; tests/test_checkpoint_cpu.py writes files in these layouts from the
synthetic weights and checks the round trip, torch weight_norm pairs included):
  gpt.pth                 torch.load(...)['model'] or the dict itself   (utils/checkpoint.py:25-36)
  s2mel.pth               state['net'][{'cfm','length_regulator','gpt_layer'}] with 'module.' stripped (commons.py:588-621);
                          weight-norm pairs (weight_g, weight_v) are folded here: w = g * v / ||v||
  bigvgan_generator.pt    ['generator'] (bigvgan.py:413-492), weight-norm folded the same way
The semantic-codec codebook/out_project for vq2emb come from the MaskGCT semantic codec safetensors (infer_v2.py:214-215).
"""
from __future__ import annotations

import os
from typing import Dict

import torch


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_v"):
            g = sd[k[:-1] + "g"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[k[: -len("_v")]] = (g / norm) * v
        elif k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out


def load_reference_checkpoints(model_dir: str):
    gpt = torch.load(os.path.join(model_dir, "gpt.pth"), map_location="cpu", weights_only=True)
    gpt = gpt.get("model", gpt)
    s2 = torch.load(os.path.join(model_dir, "s2mel.pth"), map_location="cpu", weights_only=True)["net"]
    s2mel = {}
    for sub in ("cfm", "length_regulator", "gpt_layer"):
        for k, v in fold_weight_norm({kk.replace("module.", ""): vv for kk, vv in s2[sub].items()}).items():
            s2mel[f"{sub}.{k}"] = v
    codec_path = os.path.join(model_dir, "semantic_codec.safetensors")
    if os.path.exists(codec_path):
        from safetensors.torch import load_file
        cd = fold_weight_norm(load_file(codec_path))
        for k in ("quantizer.quantizers.0.codebook.weight", "quantizer.quantizers.0.out_project.weight",
                  "quantizer.quantizers.0.out_project.bias"):
            s2mel[f"semantic_codec.{k}"] = cd[k]
    else:
        raise FileNotFoundError(codec_path)
    voc = torch.load(os.path.join(model_dir, "bigvgan_generator.pt"), map_location="cpu", weights_only=True)
    voc = fold_weight_norm(voc.get("generator", voc))
    return gpt, s2mel, voc
