"""Importer for the reference's real checkpoints (SURVEY.md §5 "Checkpoint / resume", §8f rank 4).

Layouts read (none of these files ships offline; tests/test_checkpoint_cpu.py writes files in these layouts from the
synthetic weights and checks the round trip, torch weight_norm pairs included):
  gpt.pth                 torch.load(...)['model'] or the dict itself   (utils/checkpoint.py:25-36)
  s2mel.pth               state['net'][{'cfm','length_regulator','gpt_layer'}] with 'module.' stripped (commons.py:588-621);
                          weight-norm pairs (weight_g, weight_v) are folded here: w = g * v / ||v||
  bigvgan_generator.pt    ['generator'] (bigvgan.py:413-492), weight-norm folded the same way
The semantic-codec codebook / out_project for vq2emb come from the MaskGCT semantic codec safetensors, and the vocoder from the
BigVGAN hub snapshot -- located the way the reference finds them (infer_v2.py:5, 214, 260-261: `hf_hub_download("amphion/MaskGCT",
"semantic_codec/model.safetensors")`, `BigVGAN.from_pretrained(cfg.vocoder.name)` under HF_HUB_CACHE = ./checkpoints/hf_cache),
offline: the hub CACHE is searched, nothing is fetched.  `config.yaml` (cfg_path) supplies the model dimensions.
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


def find_hub_file(repo_id: str, filename: str, model_dir: str, flat_name: str = None) -> str:
    """Path of `filename` of hub repo `repo_id` in a local hub cache -- HF_HUB_CACHE, ./checkpoints/hf_cache (what the reference
    sets at import, infer_v2.py:5), <model_dir>/hf_cache -- in the hub's own layout
    (models--org--name/snapshots/<rev>/<filename>), else a flat copy <model_dir>/<flat_name>.  Never touches the network."""
    import glob
    roots = [os.environ.get("HF_HUB_CACHE"), os.path.join(".", "checkpoints", "hf_cache"), os.path.join(model_dir, "hf_cache")]
    for root in [r for r in roots if r]:
        hits = sorted(glob.glob(os.path.join(root, "models--" + repo_id.replace("/", "--"), "snapshots", "*", filename)))
        if hits:
            return hits[-1]
    if flat_name and os.path.exists(os.path.join(model_dir, flat_name)):
        return os.path.join(model_dir, flat_name)
    raise FileNotFoundError(f"{filename} of hub repo {repo_id}: not in a local hub cache ({[r for r in roots if r]}) and no "
                            f"{flat_name} in {model_dir}; this path never downloads")


def config_from_yaml(cfg_path: str):
    """PipelineConfig from the reference's checkpoints/config.yaml (gpt: 14-44, s2mel: 53-108, vocoder: 117-119)."""
    import yaml
    from .config import BigVGANConfig, CondModuleConfig, GPTConfig, PipelineConfig, S2MelConfig
    with open(cfg_path) as f:
        y = yaml.safe_load(f)
    g, s = y["gpt"], y["s2mel"]
    cm = lambda d: CondModuleConfig(output_size=d["output_size"], linear_units=d["linear_units"], attention_heads=d["attention_heads"],
                                    num_blocks=d["num_blocks"], perceiver_mult=d["perceiver_mult"])
    gpt = GPTConfig(model_dim=g["model_dim"], heads=g["heads"], layers=g["layers"], number_text_tokens=g["number_text_tokens"],
                    number_mel_codes=g["number_mel_codes"], start_mel_token=g["start_mel_token"], stop_mel_token=g["stop_mel_token"],
                    start_text_token=g["start_text_token"], stop_text_token=g["stop_text_token"], max_mel_tokens=g["max_mel_tokens"],
                    max_text_tokens=g["max_text_tokens"], cond_module=cm(g["condition_module"]), emo_cond_module=cm(g["emo_condition_module"]))
    d, w, lr = s["DiT"], s["wavenet"], s["length_regulator"]
    s2 = S2MelConfig(hidden_dim=d["hidden_dim"], num_heads=d["num_heads"], depth=d["depth"], in_channels=d["in_channels"],
                     content_dim=d["content_dim"], style_dim=s["style_encoder"]["dim"], wn_hidden=w["hidden_dim"], wn_layers=w["num_layers"],
                     wn_kernel=w["kernel_size"], wn_dilation_rate=w["dilation_rate"], lr_channels=lr["channels"],
                     lr_in_channels=lr["in_channels"], lr_num_convs=len(lr["sampling_ratios"]), gpt_dim=g["model_dim"],
                     codebook_size=y["semantic_codec"]["codebook_size"], codebook_dim=y["semantic_codec"]["codebook_dim"],
                     codec_hidden=y["semantic_codec"]["hidden_size"])
    return PipelineConfig(gpt=gpt, s2mel=s2, bigvgan=BigVGANConfig()), y


def load_reference_checkpoints(model_dir: str, cfg: dict = None):
    """cfg: the parsed config.yaml (file names `gpt_checkpoint`, `s2mel_checkpoint`, `vocoder.name`), or None for the defaults."""
    cfg = cfg or {}
    gpt = torch.load(os.path.join(model_dir, cfg.get("gpt_checkpoint", "gpt.pth")), map_location="cpu", weights_only=True)
    gpt = gpt.get("model", gpt)
    s2 = torch.load(os.path.join(model_dir, cfg.get("s2mel_checkpoint", "s2mel.pth")), map_location="cpu", weights_only=True)["net"]
    s2mel = {}
    for sub in ("cfm", "length_regulator", "gpt_layer"):
        for k, v in fold_weight_norm({kk.replace("module.", ""): vv for kk, vv in s2[sub].items()}).items():
            s2mel[f"{sub}.{k}"] = v
    from safetensors.torch import load_file
    codec_path = find_hub_file("amphion/MaskGCT", "semantic_codec/model.safetensors", model_dir, "semantic_codec.safetensors")
    cd = fold_weight_norm(load_file(codec_path))
    for k in ("quantizer.quantizers.0.codebook.weight", "quantizer.quantizers.0.out_project.weight",
              "quantizer.quantizers.0.out_project.bias"):
        s2mel[f"semantic_codec.{k}"] = cd[k]
    voc_name = (cfg.get("vocoder") or {}).get("name", "nvidia/bigvgan_v2_22khz_80band_256x")
    voc = torch.load(find_hub_file(voc_name, "bigvgan_generator.pt", model_dir, "bigvgan_generator.pt"), map_location="cpu", weights_only=True)
    voc = fold_weight_norm(voc.get("generator", voc))
    return gpt, s2mel, voc
