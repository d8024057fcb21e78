"""GPU: `IndexTTS2(cfg_path, model_dir)` -- the reference's constructor signature (infer_v2.py:69-72) -- assembles the whole object
from a checkpoint directory in the reference's layouts (tests/ckpt_dir.py) and then `infer("voice.wav", "text", ...)` works with no
manual attachment, bit for bit like the object assembled by hand from the same state dicts."""
import os
import wave

import numpy as np
import pytest
import torch

from indextts_amd import synth

pytestmark = pytest.mark.gpu


def _audio(tag, sr, seconds):
    n = int(sr * seconds)
    t = np.arange(n) / sr
    return (0.4 * np.sin(2 * np.pi * (180 + 40 * np.sin(2 * np.pi * 1.3 * t)) * t) + 0.1 * np.sin(2 * np.pi * 1900 * t)
            + 0.05 * synth.uniform(tag, (n,), 1.0)).astype(np.float32)


def _write_wav(path, x, sr):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def test_constructor_from_checkpoint_directory(device, tmp_path):
    import warnings
    from ckpt_dir import write_checkpoint_dir
    from indextts_amd.infer_v2 import IndexTTS2, InferenceResult
    from indextts_amd.prompt import PromptEncoders
    from indextts_amd.tokenizer import TextTokenizer
    warnings.simplefilter("ignore")
    ck = tmp_path / "checkpoints"
    S = write_checkpoint_dir(ck)
    tts = IndexTTS2(cfg_path=str(ck / "config.yaml"), model_dir=str(ck), device=device)
    assert tts.cfg == S["cfg"] and tts.prompt_encoders is not None and tts.tokenizer is not None and len(tts.emo_matrix) == 3
    # the same object by hand, twice: from the state dicts the importer returns (bit for bit: the constructor adds nothing of its
    # own) and from the synthetic weights the files were written from (weight-norm pairs folded back and the statistics file's
    # sqrt(var) differ from the originals in the last bit: a tolerance)
    from indextts_amd.checkpoint import config_from_yaml, load_prompt_checkpoints, load_reference_checkpoints
    cfg2, raw = config_from_yaml(str(ck / "config.yaml"), str(ck))
    hand = IndexTTS2.from_state_dicts(cfg2, *load_reference_checkpoints(str(ck), raw), device=device)
    hand.attach_prompt_models(load_prompt_checkpoints(str(ck), raw))
    ref = IndexTTS2.from_state_dicts(S["cfg"], S["gpt"], S["s2mel"], S["voc"], device=device)
    ref.prompt_encoders = PromptEncoders(dict(S["w2v"]), S["codec"], S["campplus"], ref.s2mel, device=device, w2vbert_cfg=S["wcfg"], codec_cfg=S["ccfg"],
                                         campplus_cfg=S["pcfg"], mel_kwargs=dict(num_mels=S["cfg"].s2mel.in_channels))
    ref.tokenizer = TextTokenizer(str(ck / "bpe.model"), None)
    ref.set_emotion_matrices(S["banks"]["emo_matrix"], S["banks"]["spk_matrix"], [2, 3, 1])
    _write_wav(tmp_path / "voice.wav", _audio("t/ckpt/voice", 22050, 2.4), 22050)
    _write_wav(tmp_path / "emo.wav", _audio("t/ckpt/emo", 16000, 1.6), 16000)
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=16)
    text = "HELLO WORLD, THIS IS A TEST. AND ONE MORE SENTENCE."
    outs = []
    for obj in (tts, hand, ref):
        torch.manual_seed(11)
        r = obj.infer(str(tmp_path / "voice.wav"), text, None, emo_audio_prompt=str(tmp_path / "emo.wav"), emo_alpha=0.6, return_audio=True, **G)
        assert isinstance(r, InferenceResult) and r.sampling_rate == 22050 and r.audio.numel() > 0
        outs.append(r.audio)
    assert torch.equal(outs[0], outs[1]), "directory-constructed and hand-assembled (same loaded state dicts) objects must agree bit for bit"
    assert outs[0].shape == outs[2].shape and (outs[0] - outs[2]).abs().max().item() <= 2e-4 * 32767, "vs the object built from the original weights"
    # the file-writing form of the reference's call (infer_v2.py:905-917) and the emotion-vector mode (586-615, 668-679)
    torch.manual_seed(11)
    out = tts.infer(str(tmp_path / "voice.wav"), text, str(tmp_path / "out" / "gen.wav"), **G)
    assert out == str(tmp_path / "out" / "gen.wav") and os.path.getsize(out) > 44
    torch.manual_seed(11)
    r = tts.infer(str(tmp_path / "voice.wav"), text, None, emo_vector=[0.3, 0.0, 0.5], return_audio=True, **G)
    assert r.audio.numel() > 0 and torch.isfinite(r.audio).all()
