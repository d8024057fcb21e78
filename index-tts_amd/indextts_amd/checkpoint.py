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


def config_from_yaml(cfg_path: str, model_dir: str = None):
    """PipelineConfig from the reference's checkpoints/config.yaml (gpt: 14-44, s2mel: 53-108, vocoder: 117-119) and, for the
    vocoder, the `config.json` next to `bigvgan_generator.pt` in the hub snapshot (what `BigVGAN.from_pretrained` reads,
    bigvgan.py:413-492) when `model_dir` is given and the snapshot is there.
    Dimensions the reference hard-codes in its Python (the 32 perceiver latents, model_v2.py:416; gpt_layer 256/128/1024,
    commons.py:413; the 1024-wide conformer input, model_v2.py:396; the rotary table length) may be overridden by optional keys of
    the same sections -- `gpt.cond_latents`, `gpt.emo_perceiver_dim`, `gpt.condition_module.input_size`, `s2mel.gpt_layer_dims`,
    `s2mel.DiT.rope_block_size` -- which the reference's file does not carry: reduced-size test checkpoints use them."""
    import json
    import yaml
    from .config import BigVGANConfig, CondModuleConfig, GPTConfig, PipelineConfig, S2MelConfig
    with open(cfg_path) as f:
        y = yaml.safe_load(f)
    g, s = y["gpt"], y["s2mel"]
    cm0 = CondModuleConfig()
    cm = lambda d: CondModuleConfig(output_size=d["output_size"], linear_units=d["linear_units"], attention_heads=d["attention_heads"],
                                    num_blocks=d["num_blocks"], perceiver_mult=d["perceiver_mult"], input_size=d.get("input_size", cm0.input_size))
    g0 = GPTConfig()
    gpt = GPTConfig(model_dim=g["model_dim"], heads=g["heads"], layers=g["layers"], number_text_tokens=g["number_text_tokens"],
                    number_mel_codes=g["number_mel_codes"], start_mel_token=g["start_mel_token"], stop_mel_token=g["stop_mel_token"],
                    start_text_token=g["start_text_token"], stop_text_token=g["stop_text_token"], max_mel_tokens=g["max_mel_tokens"],
                    max_text_tokens=g["max_text_tokens"], cond_module=cm(g["condition_module"]), emo_cond_module=cm(g["emo_condition_module"]),
                    cond_latents=g.get("cond_latents", g0.cond_latents), emo_perceiver_dim=g.get("emo_perceiver_dim", g0.emo_perceiver_dim))
    d, w, lr = s["DiT"], s["wavenet"], s["length_regulator"]
    s0 = S2MelConfig()
    s2 = S2MelConfig(hidden_dim=d["hidden_dim"], num_heads=d["num_heads"], depth=d["depth"], in_channels=d["in_channels"],
                     content_dim=d["content_dim"], style_dim=s["style_encoder"]["dim"], wn_hidden=w["hidden_dim"], wn_layers=w["num_layers"],
                     wn_kernel=w["kernel_size"], wn_dilation_rate=w["dilation_rate"], lr_channels=lr["channels"],
                     lr_in_channels=lr["in_channels"], lr_num_convs=len(lr["sampling_ratios"]), gpt_dim=g["model_dim"],
                     codebook_size=y["semantic_codec"]["codebook_size"], codebook_dim=y["semantic_codec"]["codebook_dim"],
                     codec_hidden=y["semantic_codec"]["hidden_size"], gpt_layer_dims=tuple(s.get("gpt_layer_dims", s0.gpt_layer_dims)),
                     block_size=d.get("rope_block_size", s0.block_size))
    voc = BigVGANConfig()
    if model_dir is not None:
        try:
            name = (y.get("vocoder") or {}).get("name", "nvidia/bigvgan_v2_22khz_80band_256x")
            cj = os.path.join(os.path.dirname(find_hub_file(name, "bigvgan_generator.pt", model_dir, "bigvgan_generator.pt")), "config.json")
            if os.path.exists(cj):
                with open(cj) as f:
                    j = json.load(f)
                voc = BigVGANConfig(num_mels=j.get("num_mels", voc.num_mels), upsample_initial_channel=j.get("upsample_initial_channel", voc.upsample_initial_channel),
                                    upsample_rates=tuple(j.get("upsample_rates", voc.upsample_rates)),
                                    upsample_kernel_sizes=tuple(j.get("upsample_kernel_sizes", voc.upsample_kernel_sizes)),
                                    resblock_kernel_sizes=tuple(j.get("resblock_kernel_sizes", voc.resblock_kernel_sizes)),
                                    resblock_dilation_sizes=tuple(tuple(x) for x in j.get("resblock_dilation_sizes", voc.resblock_dilation_sizes)),
                                    sampling_rate=j.get("sampling_rate", voc.sampling_rate), hop_size=j.get("hop_size", voc.hop_size))
        except FileNotFoundError:
            pass
    extra = {}
    if "diffusion_steps" in y:
        extra["diffusion_steps"] = int(y["diffusion_steps"])
    return PipelineConfig(gpt=gpt, s2mel=s2, bigvgan=voc, **extra), y


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


def load_prompt_checkpoints(model_dir: str, cfg: dict = None) -> dict:
    """Everything else the reference's constructor loads (infer_v2.py:187-289), from the same places, offline:
      facebook/w2v-bert-2.0 `model.safetensors` (+ `config.json`)   hub cache            Wav2Vec2BertModel.from_pretrained (maskgct_utils.py:88)
      <model_dir>/<cfg.w2v_stat>  {"mean", "var"}                    wav2vec2bert_stats.pt  semantic_mean, semantic_std = sqrt(var) (:91-93)
      amphion/MaskGCT `semantic_codec/model.safetensors`            hub cache            the RepCodec encoder + quantizer (infer_v2.py:213-216)
      funasr/campplus `campplus_cn_common.bin`                      hub cache            CAMPPlus(feat_dim=80, embedding_size=192) (:251-257)
      <model_dir>/<cfg.emo_matrix>, <cfg.spk_matrix>, cfg.emo_num   feat2.pt / feat1.pt  the emotion banks (:281-289)
      <model_dir>/<cfg.dataset.bpe_model>                           bpe.model            TextTokenizer (:274-279)
    Returns a dict of state dicts / tensors / paths; configs of the encoders come from config.yaml (`semantic_codec`) and the
    snapshot's config.json (w2v-bert) where present, else the published defaults."""
    import json
    from safetensors.torch import load_file
    from .config import CamPPlusConfig, RepCodecConfig, W2VBertConfig
    cfg = cfg or {}
    out = {}
    w2v_path = find_hub_file("facebook/w2v-bert-2.0", "model.safetensors", model_dir, "w2v-bert-2.0.safetensors")
    out["w2vbert"] = load_file(w2v_path)
    wc = W2VBertConfig()
    cj = os.path.join(os.path.dirname(w2v_path), "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            j = json.load(f)
        wc = W2VBertConfig(input_dim=j.get("feature_projection_input_dim", wc.input_dim), hidden_size=j.get("hidden_size", wc.hidden_size),
                           num_heads=j.get("num_attention_heads", wc.num_heads), intermediate_size=j.get("intermediate_size", wc.intermediate_size),
                           num_layers=min(wc.num_layers, j.get("num_hidden_layers", wc.num_layers)),      # hidden_states[17] needs 17 layers
                           left_max=j.get("left_max_position_embeddings", wc.left_max), right_max=j.get("right_max_position_embeddings", wc.right_max),
                           conv_kernel=j.get("conv_depthwise_kernel_size", wc.conv_kernel), layer_norm_eps=j.get("layer_norm_eps", wc.layer_norm_eps))
    out["w2vbert_cfg"] = wc
    stat = torch.load(os.path.join(model_dir, cfg.get("w2v_stat", "wav2vec2bert_stats.pt")), map_location="cpu", weights_only=True)
    out["semantic_mean"], out["semantic_std"] = stat["mean"].float(), torch.sqrt(stat["var"].float())
    out["codec"] = load_file(find_hub_file("amphion/MaskGCT", "semantic_codec/model.safetensors", model_dir, "semantic_codec.safetensors"))
    sc = cfg.get("semantic_codec") or {}
    rc = RepCodecConfig()
    out["codec_cfg"] = RepCodecConfig(**{k: sc.get(k, getattr(rc, k)) for k in ("hidden_size", "codebook_size", "codebook_dim", "vocos_dim",
                                                                                 "vocos_intermediate_dim", "vocos_num_layers")})
    out["campplus"] = torch.load(find_hub_file("funasr/campplus", "campplus_cn_common.bin", model_dir, "campplus_cn_common.bin"),
                                 map_location="cpu", weights_only=True)
    cp, c0 = cfg.get("campplus") or {}, CamPPlusConfig()      # (not a section of the reference's file: CAMPPlus(feat_dim=80, embedding_size=192) is fixed there)
    out["campplus_cfg"] = CamPPlusConfig(feat_dim=cp.get("feat_dim", c0.feat_dim), embedding_size=cp.get("embedding_size", c0.embedding_size),
                                         block_layers=tuple(cp.get("block_layers", c0.block_layers)), block_dilation=tuple(cp.get("block_dilation", c0.block_dilation)))
    out["emo_matrix"] = torch.load(os.path.join(model_dir, str(cfg.get("emo_matrix", "feat2.pt")).strip()), map_location="cpu", weights_only=True)
    out["spk_matrix"] = torch.load(os.path.join(model_dir, str(cfg.get("spk_matrix", "feat1.pt")).strip()), map_location="cpu", weights_only=True)
    out["emo_num"] = list(cfg.get("emo_num", [3, 17, 2, 8, 4, 5, 10, 24]))
    out["bpe_path"] = os.path.join(model_dir, (cfg.get("dataset") or {}).get("bpe_model", "bpe.model"))
    sp = ((cfg.get("s2mel") or {}).get("preprocess_params") or {})
    spect = sp.get("spect_params") or {}
    # mel_fn_args of infer_v2.py:291-300 (fmax "None" -> None, else 8000 as the reference hard-codes; center False)
    out["mel_kwargs"] = {"n_fft": spect.get("n_fft", 1024), "win_size": spect.get("win_length", 1024), "hop_size": spect.get("hop_length", 256),
                         "num_mels": spect.get("n_mels", 80), "sampling_rate": sp.get("sr", 22050), "fmin": spect.get("fmin", 0),
                         "fmax": None if spect.get("fmax", "None") == "None" else 8000}
    return out


def reference_text_normalizer():
    """The reference's TextNormalizer (indextts/utils/front.py:11-229) when it is importable (it wraps WeTextProcessing / wetext,
    which this image lacks), else None: the tokenizer then sees the text as written."""
    try:
        from indextts.utils.front import TextNormalizer      # noqa: PLC0415 -- a maintainer's installation has it on sys.path
        return TextNormalizer()
    except Exception:                                         # noqa: BLE001
        return None
