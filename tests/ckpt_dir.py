"""Test helper: a complete IndexTTS-2 checkpoint directory in the REFERENCE's on-disk layouts, written from the synthetic weights
at reduced size -- everything `indextts.infer_v2.IndexTTS2.__init__` loads (infer_v2.py:138-289):

    config.yaml                                   the reference's sections (+ the optional size keys checkpoint.config_from_yaml documents)
    gpt.pth ['model'], s2mel.pth ['net'][...]      'module.' prefixes, weight_g / weight_v pairs as torch weight_norm stores them
    wav2vec2bert_stats.pt {'mean', 'var'}, feat1.pt, feat2.pt, bpe.model
    hf_cache/models--facebook--w2v-bert-2.0/snapshots/<rev>/{model.safetensors, config.json}
    hf_cache/models--amphion--MaskGCT/snapshots/<rev>/semantic_codec/model.safetensors
    hf_cache/models--funasr--campplus/snapshots/<rev>/campplus_cn_common.bin
    hf_cache/models--nvidia--bigvgan_v2_22khz_80band_256x/snapshots/<rev>/{bigvgan_generator.pt, config.json}
"""
import dataclasses
import json
import os
import shutil

import numpy as np
import torch

from indextts_amd import synth, weights
from indextts_amd.config import CamPPlusConfig, PipelineConfig, RepCodecConfig, W2VBertConfig

FEAT = 96          # feature width shared by the semantic model, the codec, the conformer input and the length regulator input
EMO_NUM = [2, 3, 1]


def tiny_configs():
    cfg = PipelineConfig.tiny()
    g = cfg.gpt
    g = dataclasses.replace(g, cond_module=dataclasses.replace(g.cond_module, input_size=FEAT),
                            emo_cond_module=dataclasses.replace(g.emo_cond_module, input_size=FEAT))
    cfg = dataclasses.replace(cfg, gpt=g)
    assert cfg.s2mel.lr_in_channels == FEAT
    wcfg = dataclasses.replace(W2VBertConfig.tiny(), input_dim=160, hidden_size=FEAT)
    ccfg = dataclasses.replace(RepCodecConfig.tiny(), hidden_size=FEAT, codebook_size=cfg.s2mel.codebook_size, codebook_dim=cfg.s2mel.codebook_dim)
    pcfg = dataclasses.replace(CamPPlusConfig(), embedding_size=cfg.s2mel.style_dim, block_layers=(4, 2), block_dilation=(1, 2))
    return cfg, wcfg, ccfg, pcfg


def synth_all(tag="t/ckpt"):
    """The synthetic state dicts of every model, consistent the way the real files are: the s2mel's vq2emb tensors ARE the
    semantic codec's quantizer (one safetensors file serves both, infer_v2.py:213-216 and 841-843)."""
    cfg, wcfg, ccfg, pcfg = tiny_configs()
    wg = weights.synth_gpt_weights(cfg.gpt, tag=f"{tag}/gpt")
    wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag=f"{tag}/gpt"))
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4           # fixed-length utterances
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag=f"{tag}/s2mel")
    wc = weights.synth_repcodec_weights(ccfg, tag=f"{tag}/codec")
    for k in ("codebook.weight", "out_project.weight", "out_project.bias"):
        ws[f"semantic_codec.quantizer.quantizers.0.{k}"] = wc[f"quantizer.quantizers.0.{k}"]
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag=f"{tag}/voc")
    ww = weights.synth_w2vbert_weights(wcfg, tag=f"{tag}/w2v")
    wp = weights.synth_campplus_weights(pcfg, tag=f"{tag}/campplus")
    n = sum(EMO_NUM)
    banks = {"emo_matrix": synth.uniform(f"{tag}/feat2", (n, cfg.gpt.model_dim), 0.5),
             "spk_matrix": synth.uniform(f"{tag}/feat1", (n, cfg.s2mel.style_dim), 0.5)}
    return dict(cfg=cfg, wcfg=wcfg, ccfg=ccfg, pcfg=pcfg, gpt=wg, s2mel=ws, codec=wc, voc=wv, w2v=ww, campplus=wp, banks=banks)


def _split_weight_norm(sd, pick):
    """w -> (weight_g, weight_v) with v an arbitrary positive rescale of w per dim-0 slice (what weight_norm(dim=0) stores)."""
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.asarray(v)).clone()
        if pick(k, t):
            scale = 0.5 + torch.arange(t.shape[0], dtype=torch.float32).reshape(-1, *([1] * (t.dim() - 1))) % 3
            out[k[: -len("weight")] + "weight_g"] = t.reshape(t.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (t.dim() - 1)))
            out[k[: -len("weight")] + "weight_v"] = t * scale
        else:
            out[k] = t
    return out


def reference_yaml(S) -> dict:
    cfg, ccfg, pcfg = S["cfg"], S["ccfg"], S["pcfg"]
    g, s = cfg.gpt, cfg.s2mel
    cm = lambda c: {"output_size": c.output_size, "linear_units": c.linear_units, "attention_heads": c.attention_heads, "num_blocks": c.num_blocks,
                    "input_layer": "conv2d2", "perceiver_mult": c.perceiver_mult, "input_size": c.input_size}
    return {
        "dataset": {"bpe_model": "bpe.model"},
        "gpt": {"model_dim": g.model_dim, "max_mel_tokens": g.max_mel_tokens, "max_text_tokens": g.max_text_tokens, "heads": g.heads, "layers": g.layers,
                "number_text_tokens": g.number_text_tokens, "number_mel_codes": g.number_mel_codes, "start_mel_token": g.start_mel_token,
                "stop_mel_token": g.stop_mel_token, "start_text_token": g.start_text_token, "stop_text_token": g.stop_text_token,
                "condition_type": "conformer_perceiver", "condition_module": cm(g.cond_module), "emo_condition_module": cm(g.emo_cond_module),
                "cond_latents": g.cond_latents, "emo_perceiver_dim": g.emo_perceiver_dim},
        "semantic_codec": {"codebook_size": ccfg.codebook_size, "hidden_size": ccfg.hidden_size, "codebook_dim": ccfg.codebook_dim, "vocos_dim": ccfg.vocos_dim,
                           "vocos_intermediate_dim": ccfg.vocos_intermediate_dim, "vocos_num_layers": ccfg.vocos_num_layers},
        "s2mel": {"preprocess_params": {"sr": 22050, "spect_params": {"n_fft": 1024, "win_length": 1024, "hop_length": 256, "n_mels": s.in_channels, "fmin": 0, "fmax": "None"}},
                  "style_encoder": {"dim": s.style_dim},
                  "length_regulator": {"channels": s.lr_channels, "in_channels": s.lr_in_channels, "sampling_ratios": [1] * s.lr_num_convs},
                  "DiT": {"hidden_dim": s.hidden_dim, "num_heads": s.num_heads, "depth": s.depth, "in_channels": s.in_channels, "content_dim": s.content_dim,
                          "block_size": 8192, "rope_block_size": s.block_size},
                  "wavenet": {"hidden_dim": s.wn_hidden, "num_layers": s.wn_layers, "kernel_size": s.wn_kernel, "dilation_rate": s.wn_dilation_rate},
                  "gpt_layer_dims": list(s.gpt_layer_dims)},
        "campplus": {"embedding_size": pcfg.embedding_size, "block_layers": list(pcfg.block_layers), "block_dilation": list(pcfg.block_dilation)},
        "gpt_checkpoint": "gpt.pth", "w2v_stat": "wav2vec2bert_stats.pt", "s2mel_checkpoint": "s2mel.pth", "emo_matrix": "feat2.pt ", "spk_matrix": "feat1.pt",
        "emo_num": list(EMO_NUM), "qwen_emo_path": "qwen0.6bemo4-merge/", "vocoder": {"type": "bigvgan", "name": "nvidia/bigvgan_v2_22khz_80band_256x"},
        "version": 2.0, "diffusion_steps": cfg.diffusion_steps,
    }


def write_checkpoint_dir(root, S=None) -> dict:
    """Writes the directory under `root` (a path); returns the synthetic state dicts it was written from (synth_all())."""
    import yaml
    from safetensors.torch import save_file
    S = S or synth_all()
    root = str(root)
    os.makedirs(root, exist_ok=True)
    cfg, wcfg = S["cfg"], S["wcfg"]
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)).clone() for k, v in d.items()}
    with open(os.path.join(root, "config.yaml"), "w") as f:
        yaml.safe_dump(reference_yaml(S), f)
    torch.save({"model": tt(S["gpt"])}, os.path.join(root, "gpt.pth"))
    net = {"cfm": {}, "length_regulator": {}, "gpt_layer": {}}
    for k, v in S["s2mel"].items():
        head, rest = k.split(".", 1)
        if head in net:
            net[head][rest] = v
    conv3 = lambda k, t: k.endswith(".weight") and t.dim() == 3
    net = {sub: {"module." + k: v for k, v in _split_weight_norm(sd, lambda k, t: conv3(k, t) and "wavenet" in k).items()} for sub, sd in net.items()}
    torch.save({"net": net}, os.path.join(root, "s2mel.pth"))
    # wav2vec2bert_stats.pt holds the VARIANCE (maskgct_utils.py:92 takes its square root)
    w2v = dict(S["w2v"])
    mean, std = w2v.pop("semantic_mean"), w2v.pop("semantic_std")
    torch.save({"mean": torch.from_numpy(mean), "var": torch.from_numpy(std.astype(np.float64) ** 2).float()}, os.path.join(root, "wav2vec2bert_stats.pt"))
    torch.save(torch.from_numpy(S["banks"]["emo_matrix"]), os.path.join(root, "feat2.pt"))
    torch.save(torch.from_numpy(S["banks"]["spk_matrix"]), os.path.join(root, "feat1.pt"))
    shutil.copy(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_bpe.model"), os.path.join(root, "bpe.model"))
    hub = os.path.join(root, "hf_cache")

    def snap(repo, rev):
        p = os.path.join(hub, "models--" + repo.replace("/", "--"), "snapshots", rev)
        os.makedirs(p, exist_ok=True)
        return p
    p = snap("facebook/w2v-bert-2.0", "da985ba0")
    save_file(tt(w2v), os.path.join(p, "model.safetensors"))
    with open(os.path.join(p, "config.json"), "w") as f:
        json.dump({"feature_projection_input_dim": wcfg.input_dim, "hidden_size": wcfg.hidden_size, "num_attention_heads": wcfg.num_heads,
                   "intermediate_size": wcfg.intermediate_size, "num_hidden_layers": wcfg.num_layers, "left_max_position_embeddings": wcfg.left_max,
                   "right_max_position_embeddings": wcfg.right_max, "conv_depthwise_kernel_size": wcfg.conv_kernel, "layer_norm_eps": wcfg.layer_norm_eps,
                   "model_type": "wav2vec2-bert"}, f)
    p = snap("amphion/MaskGCT", "0123abcd")
    os.makedirs(os.path.join(p, "semantic_codec"), exist_ok=True)
    save_file(tt(S["codec"]), os.path.join(p, "semantic_codec", "model.safetensors"))
    p = snap("funasr/campplus", "fe4f9a1c")
    torch.save(tt(S["campplus"]), os.path.join(p, "campplus_cn_common.bin"))
    p = snap("nvidia/bigvgan_v2_22khz_80band_256x", "fedc9876")
    torch.save({"generator": _split_weight_norm(S["voc"], conv3)}, os.path.join(p, "bigvgan_generator.pt"))
    v = cfg.bigvgan
    with open(os.path.join(p, "config.json"), "w") as f:
        json.dump({"num_mels": v.num_mels, "upsample_initial_channel": v.upsample_initial_channel, "upsample_rates": list(v.upsample_rates),
                   "upsample_kernel_sizes": list(v.upsample_kernel_sizes), "resblock_kernel_sizes": list(v.resblock_kernel_sizes),
                   "resblock_dilation_sizes": [list(x) for x in v.resblock_dilation_sizes], "sampling_rate": v.sampling_rate, "hop_size": v.hop_size,
                   "resblock": "1", "activation": "snakebeta", "snake_logscale": True}, f)
    return S
