"""CPU: the real-checkpoint importer on files written in the reference's on-disk layouts
(gpt.pth ['model'], s2mel.pth ['net'][...] with 'module.' prefixes, bigvgan_generator.pt ['generator'], weight-norm pairs
weight_g / weight_v as torch.nn.utils.weight_norm stores them) round-trips to the state dicts the HIP contexts consume."""
import os

import numpy as np
import torch

from indextts_amd import weights
from indextts_amd.checkpoint import fold_weight_norm, load_reference_checkpoints
from indextts_amd.config import PipelineConfig


def _split_weight_norm(sd, pick):
    """w -> (weight_g, weight_v) with v an arbitrary positive rescale of w per dim-0 slice (what weight_norm(dim=0) stores)."""
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        t = torch.from_numpy(np.asarray(v)).clone()
        if pick(k, t):
            scale = 0.5 + torch.arange(t.shape[0], dtype=torch.float32).reshape(-1, *([1] * (t.dim() - 1))) % 3
            wv = t * scale
            out[k[: -len("weight")] + "weight_g"] = t.reshape(t.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (t.dim() - 1)))
            out[k[: -len("weight")] + "weight_v"] = wv
        else:
            out[k] = t
    return out


def test_reference_checkpoint_layouts_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/ckpt/gpt")
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="t/ckpt/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/ckpt/voc")
    torch.save({"model": {k: torch.from_numpy(v) for k, v in wg.items()}}, tmp_path / "gpt.pth")
    net = {"cfm": {}, "length_regulator": {}, "gpt_layer": {}}
    codec = {}
    for k, v in ws.items():
        head, rest = k.split(".", 1)
        if head == "semantic_codec":
            codec[rest] = torch.from_numpy(v).clone()
        elif head in net:
            net[head][rest] = v
    conv3 = lambda k, t: k.endswith(".weight") and t.dim() == 3
    net = {sub: {"module." + k: v for k, v in _split_weight_norm(sd, lambda k, t: conv3(k, t) and "wavenet" in k).items()}
           for sub, sd in net.items()}
    assert any(k.endswith("weight_g") for k in net["cfm"]), "fixture should contain weight-normed WaveNet layers"
    torch.save({"net": net}, tmp_path / "s2mel.pth")
    # the semantic codec and the vocoder in a hub CACHE laid out as huggingface_hub lays it out (the reference finds them with
    # hf_hub_download / from_pretrained under HF_HUB_CACHE, infer_v2.py:5, 214, 260); nothing is downloaded
    snap = tmp_path / "hf_cache" / "models--amphion--MaskGCT" / "snapshots" / "0123abcd" / "semantic_codec"
    snap.mkdir(parents=True)
    save_file(codec, str(snap / "model.safetensors"))
    vsnap = tmp_path / "hf_cache" / "models--nvidia--bigvgan_v2_22khz_80band_256x" / "snapshots" / "fedc9876"
    vsnap.mkdir(parents=True)
    torch.save({"generator": _split_weight_norm(wv, conv3)}, vsnap / "bigvgan_generator.pt")

    gpt, s2mel, voc = load_reference_checkpoints(str(tmp_path))
    for got, want, name in ((gpt, wg, "gpt"), (s2mel, ws, "s2mel"), (voc, wv, "bigvgan")):
        assert set(got) == set(want), (name, sorted(set(got) ^ set(want))[:5])
        for k in want:
            np.testing.assert_allclose(got[k].numpy(), np.asarray(want[k]), rtol=2e-6, atol=1e-7, err_msg=f"{name}:{k}")


def test_fold_weight_norm_matches_torch():
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 10, 5))
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 8, stride=4))
    for m in (conv, convt):
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        assert any(k.endswith("weight_g") for k in sd)
        folded = fold_weight_norm(sd)
        assert torch.allclose(folded["weight"], m.weight.detach(), atol=1e-6)


def test_config_from_reference_yaml(tmp_path):
    """cfg_path: the dimensions come from checkpoints/config.yaml (its gpt / s2mel / semantic_codec sections); written here with the
    values of the reference's file, which equal PipelineConfig's defaults."""
    import yaml
    from indextts_amd.checkpoint import config_from_yaml
    ref = {"gpt": {"model_dim": 1280, "max_mel_tokens": 1815, "max_text_tokens": 600, "heads": 20, "layers": 24, "number_text_tokens": 12000,
                   "number_mel_codes": 8194, "start_mel_token": 8192, "stop_mel_token": 8193, "start_text_token": 0, "stop_text_token": 1,
                   "condition_module": {"output_size": 512, "linear_units": 2048, "attention_heads": 8, "num_blocks": 6, "input_layer": "conv2d2", "perceiver_mult": 2},
                   "emo_condition_module": {"output_size": 512, "linear_units": 1024, "attention_heads": 4, "num_blocks": 4, "input_layer": "conv2d2", "perceiver_mult": 2}},
           "semantic_codec": {"codebook_size": 8192, "hidden_size": 1024, "codebook_dim": 8},
           "s2mel": {"style_encoder": {"dim": 192},
                     "length_regulator": {"channels": 512, "in_channels": 1024, "sampling_ratios": [1, 1, 1, 1]},
                     "DiT": {"hidden_dim": 512, "num_heads": 8, "depth": 13, "in_channels": 80, "content_dim": 512},
                     "wavenet": {"hidden_dim": 512, "num_layers": 8, "kernel_size": 5, "dilation_rate": 1}},
           "gpt_checkpoint": "gpt.pth", "s2mel_checkpoint": "s2mel.pth", "vocoder": {"type": "bigvgan", "name": "nvidia/bigvgan_v2_22khz_80band_256x"}}
    path = tmp_path / "config.yaml"
    path.write_text(yaml.safe_dump(ref))
    cfg, raw = config_from_yaml(str(path))
    want = PipelineConfig()
    assert cfg.gpt == want.gpt and cfg.s2mel.hidden_dim == want.s2mel.hidden_dim and cfg.s2mel.ffn_dim == want.s2mel.ffn_dim
    assert cfg.s2mel.lr_num_convs == 4 and cfg.s2mel.codebook_size == 8192 and raw["vocoder"]["name"].startswith("nvidia/")
    ref["gpt"]["layers"] = 12
    path.write_text(yaml.safe_dump(ref))
    assert config_from_yaml(str(path))[0].gpt.layers == 12


def test_prompt_side_checkpoints_round_trip(tmp_path):
    """Everything else the reference's constructor loads (infer_v2.py:187-289: w2v-bert snapshot + statistics, semantic codec,
    CAMPPlus, emotion banks, bpe.model), written by tests/ckpt_dir.py in the reference's layouts, comes back as the state dicts the
    HIP contexts consume -- and the dimensions come back from config.yaml / the snapshots' config.json."""
    from ckpt_dir import EMO_NUM, write_checkpoint_dir
    from indextts_amd.checkpoint import config_from_yaml, load_prompt_checkpoints
    S = write_checkpoint_dir(tmp_path)
    cfg, raw = config_from_yaml(str(tmp_path / "config.yaml"), str(tmp_path))
    assert cfg == S["cfg"], "config.yaml (+ BigVGAN config.json) must reproduce the pipeline configuration"
    gpt, s2mel, voc = load_reference_checkpoints(str(tmp_path), raw)
    for got, want, name in ((gpt, S["gpt"], "gpt"), (s2mel, S["s2mel"], "s2mel"), (voc, S["voc"], "bigvgan")):
        assert set(got) == set(want), (name, sorted(set(got) ^ set(want))[:5])
        for k in want:
            np.testing.assert_allclose(got[k].numpy(), np.asarray(want[k]), rtol=2e-6, atol=1e-7, err_msg=f"{name}:{k}")
    ck = load_prompt_checkpoints(str(tmp_path), raw)
    assert ck["w2vbert_cfg"] == S["wcfg"] and ck["codec_cfg"] == S["ccfg"] and ck["campplus_cfg"] == S["pcfg"]
    w2v = dict(S["w2v"])
    mean, std = w2v.pop("semantic_mean"), w2v.pop("semantic_std")
    np.testing.assert_array_equal(ck["semantic_mean"].numpy(), mean)
    np.testing.assert_allclose(ck["semantic_std"].numpy(), std, rtol=2e-7)        # stored as the variance, as the reference's file does
    for got, want, name in ((ck["w2vbert"], w2v, "w2vbert"), (ck["codec"], S["codec"], "codec"), (ck["campplus"], S["campplus"], "campplus")):
        assert set(got) == set(want), (name, sorted(set(got) ^ set(want))[:5])
        for k in want:
            np.testing.assert_array_equal(got[k].numpy(), np.asarray(want[k]), err_msg=f"{name}:{k}")
    np.testing.assert_array_equal(ck["emo_matrix"].numpy(), S["banks"]["emo_matrix"])
    np.testing.assert_array_equal(ck["spk_matrix"].numpy(), S["banks"]["spk_matrix"])
    assert ck["emo_num"] == EMO_NUM and os.path.exists(ck["bpe_path"]) and ck["mel_kwargs"]["num_mels"] == cfg.s2mel.in_channels
    assert ck["mel_kwargs"]["fmax"] is None


def test_missing_prompt_side_file_is_reported(tmp_path):
    import pytest
    from ckpt_dir import write_checkpoint_dir
    from indextts_amd.checkpoint import load_prompt_checkpoints
    write_checkpoint_dir(tmp_path)
    os.remove(tmp_path / "feat1.pt")
    with pytest.raises(FileNotFoundError):
        load_prompt_checkpoints(str(tmp_path), {"spk_matrix": "feat1.pt"})
