"""GPU: the audio-side prompt block end to end (indextts_amd/prompt.py: features -> w2v-bert layers -> semantic codec -> length regulator;
mel; fbank -> CAMPPlus), stage by stage against the CPU oracles, and `infer()` taking the resampled prompt audio itself."""
import dataclasses

import numpy as np
import pytest
import torch

from indextts_amd import features, synth, weights
from indextts_amd.config import CamPPlusConfig, PipelineConfig, RepCodecConfig, W2VBertConfig

pytestmark = pytest.mark.gpu

FEAT = 96          # feature width shared by the semantic model, the codec, the conformer input and the length regulator input


def _cfgs():
    cfg = PipelineConfig.tiny()
    g = cfg.gpt
    g = dataclasses.replace(g, cond_module=dataclasses.replace(g.cond_module, input_size=FEAT),
                            emo_cond_module=dataclasses.replace(g.emo_cond_module, input_size=FEAT))
    cfg = dataclasses.replace(cfg, gpt=g)
    assert cfg.s2mel.lr_in_channels == FEAT
    wcfg = dataclasses.replace(W2VBertConfig.tiny(), input_dim=160, hidden_size=FEAT)
    ccfg = dataclasses.replace(RepCodecConfig.tiny(), hidden_size=FEAT)
    pcfg = dataclasses.replace(CamPPlusConfig(), embedding_size=cfg.s2mel.style_dim, block_layers=(4, 2), block_dilation=(1, 2))
    return cfg, wcfg, ccfg, pcfg


def _audio(tag, sr, seconds):
    n = int(sr * seconds)
    t = np.arange(n) / sr
    return (0.4 * np.sin(2 * np.pi * (180 + 40 * np.sin(2 * np.pi * 1.3 * t)) * t) + 0.1 * np.sin(2 * np.pi * 1900 * t)
            + 0.05 * synth.uniform(tag, (n,), 1.0)).astype(np.float32)


@pytest.fixture(scope="module")
def rig(device):
    from indextts_amd.infer_v2 import IndexTTS2
    from indextts_amd.prompt import PromptEncoders
    cfg, wcfg, ccfg, pcfg = _cfgs()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/prompt/gpt")
    wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag="t/prompt/gpt"))
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4           # fixed-length utterances
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="t/prompt/s2mel")
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/prompt/voc"), device=device)
    ww = weights.synth_w2vbert_weights(wcfg, tag="t/prompt/w2v")
    wc = weights.synth_repcodec_weights(ccfg, tag="t/prompt/codec")
    wp = weights.synth_campplus_weights(pcfg, tag="t/prompt/campplus")
    enc = PromptEncoders(ww, wc, wp, tts.s2mel, device=device, w2vbert_cfg=wcfg, codec_cfg=ccfg, campplus_cfg=pcfg,
                         mel_kwargs=dict(num_mels=cfg.s2mel.in_channels))
    return cfg, wcfg, ccfg, pcfg, ws, ww, wc, wp, tts, enc


def test_encode_stage_by_stage_vs_oracles(device, rig):
    from indextts_amd.audio import slaney_mel_basis
    from indextts_amd.prompt import PromptAudio
    from oracle import audio as oa, campplus as ocp, codec as ocd, s2mel as osm, semantic as osem
    cfg, wcfg, ccfg, pcfg, ws, ww, wc, wp, tts, enc = rig
    a16, a22 = _audio("t/prompt/a16", 16000, 2.6), _audio("t/prompt/a22", 22050, 2.6)
    got = enc.encode(PromptAudio(a16, a22), PromptAudio(_audio("t/prompt/e16", 16000, 1.7)))
    tw = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    with torch.no_grad():
        f = features.seamless_m4t_features(a16)
        n = int(f["attention_mask"].sum())
        emb = osem.get_emb(tw(ww), wcfg, torch.from_numpy(f["input_features"]), torch.from_numpy(f["attention_mask"]))[:, :n]
        assert got.spk_cond_emb.shape == emb.shape and (got.spk_cond_emb.cpu() - emb).abs().max().item() <= 2e-4
        fe = features.seamless_m4t_features(_audio("t/prompt/e16", 16000, 1.7))
        ne = int(fe["attention_mask"].sum())
        emo = osem.get_emb(tw(ww), wcfg, torch.from_numpy(fe["input_features"]), torch.from_numpy(fe["attention_mask"]))[:, :ne]
        assert (got.emo_cond_emb.cpu() - emo).abs().max().item() <= 2e-4
        # the two prompts went through w2v-bert as ONE ragged batch: each row is its own B = 1 call (at this toy size the batch has 260
        # GEMM rows and a prompt alone 130: the batch is on the split-bf16 GEMM, the single call below 256 rows on the exact one --
        # a real prompt of more than 5 s is on the split-bf16 kernel either way)
        for one, solo in ((got.spk_cond_emb, enc.get_emb(a16)), (got.emo_cond_emb, enc.get_emb(_audio("t/prompt/e16", 16000, 1.7)))):
            assert one.shape == solo.shape and (one - solo).abs().max().item() <= 2e-4
        # The semantic codes of the DEFAULT arithmetic mode (w2v-bert linears on the split-bf16 GEMM from 256 rows up) against the oracle's codes
        # on the oracle's own features -- the whole chain, nothing handed over: equal, or a frame differs where the ORACLE's distances to
        # its code and to this path's code are within 1e-3 of each other (unit vectors: distances in [0, 4]), on at most 3 % of the frames
        import torch.nn.functional as F
        codes_hip, _ = enc.codec.quantize(got.spk_cond_emb)
        cw = tw(wc)
        idx_ref, _ = ocd.quantize(cw, emb)
        codes_hip = codes_hip.cpu().reshape(idx_ref.shape)
        diff = (codes_hip != idx_ref).reshape(-1).nonzero().reshape(-1)
        assert diff.numel() <= max(1, int(0.03 * idx_ref.numel())), f"{diff.numel()} of {idx_ref.numel()} semantic codes differ"
        if diff.numel():
            q = "quantizer.quantizers.0"
            z_e = F.conv1d(ocd.encoder(cw, emb).transpose(1, 2), cw[q + ".in_project.weight"], cw[q + ".in_project.bias"])
            e = F.normalize(z_e.transpose(1, 2).reshape(-1, z_e.shape[1]))
            cb = F.normalize(cw[q + ".codebook.weight"])
            for t in diff.tolist():
                d_ref = (e[t] - cb[idx_ref.reshape(-1)[t]]).pow(2).sum().item()
                d_hip = (e[t] - cb[codes_hip.reshape(-1)[t]]).pow(2).sum().item()
                assert 0.0 <= d_hip - d_ref <= 1e-3, f"frame {t}: the oracle's code is closer by {d_hip - d_ref:.3e}"
        # the next stage's oracle gets THIS path's features, so a near-tie of the nearest-code search upstream cannot cascade
        _, S_ref = ocd.quantize(tw(wc), got.spk_cond_emb.cpu())
        mel = oa.mel_spectrogram(torch.from_numpy(a22[None]), torch.from_numpy(slaney_mel_basis(22050, 1024, cfg.s2mel.in_channels)))
        assert got.ref_mel.shape == mel.shape and (got.ref_mel.cpu() - mel).abs().max().item() <= 2e-3
        pc = osm.length_regulator(tw(ws), cfg.s2mel, S_ref, torch.LongTensor([mel.shape[2]]))
        assert got.prompt_condition.shape == pc.shape
        assert (got.prompt_condition.cpu() - pc).abs().max().item() <= 2e-4 * max(1.0, pc.abs().max().item())
        fb = features.kaldi_fbank(a16)
        fb = fb - fb.mean(0, keepdims=True)
        style = ocp.forward(tw(wp), pcfg, torch.from_numpy(fb[None]))
        assert got.style.shape == style.shape == (1, cfg.s2mel.style_dim)
        assert (got.style.cpu() - style).abs().max().item() <= 5e-4 * max(1.0, style.abs().max().item())


def test_infer_takes_the_prompt_audio(device, rig):
    from indextts_amd.prompt import PromptAudio
    cfg, wcfg, ccfg, pcfg, ws, ww, wc, wp, tts, enc = rig
    spk = PromptAudio(_audio("t/prompt/a16", 16000, 2.6), _audio("t/prompt/a22", 22050, 2.6))
    emo = PromptAudio(_audio("t/prompt/e16", 16000, 1.7))
    seg = synth.integers("t/prompt/seg", (2, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=16)
    import warnings
    warnings.simplefilter("ignore")
    with pytest.raises(RuntimeError):
        tts.infer(spk, seg, None, **G)                       # encoders not attached yet
    tts.prompt_encoders = enc
    torch.manual_seed(3)
    sr, a = tts.infer(spk, seg, None, emo_audio_prompt=emo, emo_alpha=0.7, **G)
    cached = tts._audio_cache
    torch.manual_seed(3)
    _, a2 = tts.infer(spk, seg, None, emo_audio_prompt=emo, emo_alpha=0.7, **G)
    assert tts._audio_cache is cached and np.array_equal(a, a2)          # the prompt block ran once
    torch.manual_seed(3)
    _, b = tts.infer(enc.encode(spk, emo), seg, None, emo_alpha=0.7, **G)
    assert sr == 22050 and a.dtype == np.int16 and np.array_equal(a, b)



def test_infer_takes_wav_file_paths(device, rig, tmp_path):
    """`infer(spk_audio_prompt="speaker.wav", emo_audio_prompt="emotion.wav")` as the reference is called (infer_v2.py:628-630, 685):
    the files are read, cut and resampled on the host (indextts_amd/audioio.py) and give the same waveform as the PromptAudio built
    from the same samples by hand."""
    import wave
    from indextts_amd import audioio
    from indextts_amd.prompt import PromptAudio
    cfg, wcfg, ccfg, pcfg, ws, ww, wc, wp, tts, enc = rig
    tts.prompt_encoders = enc

    def write(path, x, sr):
        with wave.open(str(path), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
            w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())

    write(tmp_path / "spk.wav", _audio("t/prompt/f22", 22050, 2.4), 22050)      # at librosa's default rate: no first-stage resampling
    write(tmp_path / "emo.wav", _audio("t/prompt/f24", 24000, 1.5), 24000)      # resampled to 16 kHz by the loader
    seg = synth.integers("t/prompt/seg", (2, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=16)
    import warnings
    warnings.simplefilter("ignore")
    torch.manual_seed(5)
    sr, a = tts.infer(str(tmp_path / "spk.wav"), seg, None, emo_audio_prompt=str(tmp_path / "emo.wav"), emo_alpha=0.6, **G)
    files = dict(tts._prompt_files)
    torch.manual_seed(5)
    _, a2 = tts.infer(str(tmp_path / "spk.wav"), seg, None, emo_audio_prompt=str(tmp_path / "emo.wav"), emo_alpha=0.6, **G)
    assert all(tts._prompt_files[k] is v for k, v in files.items()) and np.array_equal(a, a2)      # each file was read once
    x22, r = audioio.load_and_cut_audio(str(tmp_path / "spk.wav"), 15)
    assert r == 22050
    spk = PromptAudio(audioio.sinc_resample(x22, 22050, 16000)[0], x22[0])
    emo = PromptAudio(audioio.load_and_cut_audio(str(tmp_path / "emo.wav"), 15, sr=16000)[0][0])
    torch.manual_seed(5)
    _, b = tts.infer(spk, seg, None, emo_audio_prompt=emo, emo_alpha=0.6, **G)
    assert sr == 22050 and a.dtype == np.int16 and np.array_equal(a, b)
    # one cached file per kind (the reference's cache_spk_audio_prompt / cache_emo_audio_prompt, infer_v2.py:304-310): another path replaces
    # the entry, and the same path is read again once the file has changed
    assert set(tts._prompt_files) == {"spk", "emo"}
    write(tmp_path / "spk2.wav", _audio("t/prompt/f22b", 22050, 2.0), 22050)
    torch.manual_seed(5)
    _, c = tts.infer(str(tmp_path / "spk2.wav"), seg, None, emo_audio_prompt=str(tmp_path / "emo.wav"), emo_alpha=0.6, **G)
    assert set(tts._prompt_files) == {"spk", "emo"} and tts._prompt_files["spk"][0][0].endswith("spk2.wav") and not np.array_equal(c, a)
    import os, time
    write(tmp_path / "spk2.wav", _audio("t/prompt/f22", 22050, 2.4), 22050)      # the first speaker's samples under the second name
    os.utime(tmp_path / "spk2.wav", ns=(time.time_ns(), time.time_ns() + 10_000_000))
    torch.manual_seed(5)
    _, d = tts.infer(str(tmp_path / "spk2.wav"), seg, None, emo_audio_prompt=str(tmp_path / "emo.wav"), emo_alpha=0.6, **G)
    assert np.array_equal(d, a)
