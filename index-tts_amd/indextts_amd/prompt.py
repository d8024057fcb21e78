"""The audio-side half of the reference's prompt block (infer_v2.py:618-660, 677-694) on the HIP path.

    enc = PromptEncoders(w2vbert_sd, codec_sd, campplus_sd, tts.s2mel)      # state dicts in the reference modules' own key layouts
    feats = enc.encode(PromptAudio(audio_16k, audio_22k))                   # -> PromptFeatures (spk_cond_emb, style, prompt_condition, ref_mel)
    tts.infer(feats, token_segments, None, return_audio=True)               # or tts.prompt_encoders = enc; tts.infer(PromptAudio(...), ...)

What stays with the caller: reading the file, cutting it to 15 s and resampling it to 16 kHz and 22.05 kHz (librosa.load /
torchaudio.transforms.Resample in the reference, infer_v2.py:628-631 -- neither library is in this image, so there is nothing to pin a
resampler against); `PromptAudio` carries the two resampled waveforms.  Per step:
    extract_features (SeamlessM4TFeatureExtractor)   -> features.seamless_m4t_features        host numpy, as in the reference
    get_emb (w2v-bert-2.0, hidden_states[17], stats) -> semantic.SemanticModel                 csrc/semantic.hip
    semantic_codec.quantize                          -> codec.SemanticCodec                    csrc/codec.hip
    mel_fn                                           -> audio.MelSpectrogram                   csrc/audio.hip
    kaldi.fbank - mean, campplus_model               -> features.kaldi_fbank, campplus.CAMPPlus  csrc/campplus.hip
    length_regulator(S_ref, ylens=[ref_mel frames])  -> S2Mel.length_regulator                 csrc/s2mel.hip
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import features
from .audio import MelSpectrogram
from .campplus import CAMPPlus
from .codec import SemanticCodec
from .config import CamPPlusConfig, RepCodecConfig, W2VBertConfig
from .semantic import SemanticModel


@dataclass
class PromptAudio:
    """One prompt, already cut (<= 15 s, infer_v2.py:628) and resampled: mono float waveforms in [-1, 1]."""
    audio_16k: np.ndarray
    audio_22k: Optional[np.ndarray] = None       # not needed for an emotion prompt (infer_v2.py:678-686 uses the 16 kHz audio only)


class PromptEncoders:
    def __init__(self, w2vbert_sd, codec_sd, campplus_sd, s2mel, device="cuda:0", w2vbert_cfg: W2VBertConfig = W2VBertConfig(),
                 codec_cfg: RepCodecConfig = RepCodecConfig(), campplus_cfg: CamPPlusConfig = CamPPlusConfig(), semantic_mean=None,
                 semantic_std=None, mel_kwargs: Optional[dict] = None):
        self.device = torch.device(device)
        self.semantic = SemanticModel(w2vbert_sd, w2vbert_cfg, device=self.device, mean=semantic_mean, std=semantic_std)
        self.codec = SemanticCodec(codec_sd, codec_cfg, device=self.device)
        self.campplus = CAMPPlus(campplus_sd, campplus_cfg, device=self.device)
        self.mel = MelSpectrogram(device=self.device, **(mel_kwargs or {}))
        self.s2mel = s2mel

    def get_emb(self, audio_16k) -> torch.Tensor:
        """extract_features + get_emb (infer_v2.py:633-638, 680-686) -> [1, T, 1024] (valid frames only)."""
        f = features.seamless_m4t_features(np.asarray(audio_16k, np.float32))
        emb = self.semantic(torch.from_numpy(f["input_features"]), torch.from_numpy(f["attention_mask"]))
        return emb[:, : int(f["attention_mask"].sum())]

    def get_emb_batch(self, audios_16k) -> list:
        """get_emb of several prompts as ONE ragged batch (right-padded, attention-masked): the speaker and the emotion prompt of a
        request go through the 17 w2v-bert layers together -- twice the GEMM rows per launch (a 15 s prompt is 750 frames: 6 row tiles
        of 128 under-fill 256 CUs), half the launches.  Each row equals its own B = 1 call (rows and valid frames are independent in
        every kernel; tests/test_prompt_gpu.py)."""
        fs = [features.seamless_m4t_features(np.asarray(a, np.float32)) for a in audios_16k]
        lens = [int(f["attention_mask"].sum()) for f in fs]
        T = max(f["input_features"].shape[1] for f in fs)
        x = np.zeros((len(fs), T, fs[0]["input_features"].shape[2]), np.float32)
        m = np.zeros((len(fs), T), np.int64)
        for i, f in enumerate(fs):
            t = f["input_features"].shape[1]
            x[i, :t] = f["input_features"][0]
            m[i, :lens[i]] = 1
        emb = self.semantic(torch.from_numpy(x), torch.from_numpy(m))
        return [emb[i:i + 1, :lens[i]].contiguous() for i in range(len(fs))]

    def encode(self, prompt: PromptAudio, emo_prompt: Optional[PromptAudio] = None):
        from .infer_v2 import PromptFeatures
        if prompt.audio_22k is None:
            raise ValueError("the speaker prompt needs its 22.05 kHz waveform (ref_mel)")
        a16 = np.asarray(prompt.audio_16k, np.float32).reshape(-1)
        emo = None
        if emo_prompt is None:
            spk_cond_emb = self.get_emb(a16)
        else:
            spk_cond_emb, emo = self.get_emb_batch([a16, np.asarray(emo_prompt.audio_16k, np.float32).reshape(-1)])
        _, S_ref = self.codec.quantize(spk_cond_emb)                                          # infer_v2.py:637
        ref_mel = self.mel(torch.from_numpy(np.asarray(prompt.audio_22k, np.float32).reshape(1, -1)).to(self.device))     # 640
        feat = features.kaldi_fbank(a16)                                                      # 642-645 (dither 0, 80 bins)
        feat = feat - feat.mean(axis=0, keepdims=True)                                        # 646
        style = self.campplus(torch.from_numpy(feat[None]))                                   # 647
        prompt_condition = self.s2mel.length_regulator(S_ref, ylens=torch.LongTensor([ref_mel.size(2)]), n_quantizers=3, f0=None)[0]   # 649-652
        return PromptFeatures(spk_cond_emb, style, prompt_condition, ref_mel, emo)
