"""Prompt audio from a file: read, cut, resample (host side of `infer_v2.py:510-523, 628-630, 685`).

The reference does this with two third-party packages that are absent from this image (so this module is **parity-unpinned**;
what can be checked without them is checked in `tests/test_audioio_cpu.py`):

* `librosa.load(path)` (librosa 0.10.2.post1 in the reference's lock file): decode, average the channels (`to_mono`), resample to
  22 050 Hz -- or to 16 000 Hz for the emotion prompt (`sr=16000`) -- with `soxr_hq`.  Here: RIFF/WAVE files through the standard
  library (`wave`: PCM 8 / 16 / 24 / 32 bit; IEEE float 32 is parsed by hand), channel mean, and for a file whose rate differs from the
  target the windowed-sinc resampler below in place of soxr (a different low-pass: band-limited to 0.99 of Nyquist with a Hann window
  instead of soxr's polyphase FIR; identical when the file already has the target rate, which is the documented recommendation).
* `torchaudio.transforms.Resample(sr, 16000)` / `(sr, 22050)` (torchaudio 2.8 there; defaults `sinc_interp_hann`,
  `lowpass_filter_width=6`, `rolloff=0.99`): restated in `sinc_resample` from torchaudio's published algorithm
  (`torchaudio/functional/functional.py::_get_sinc_resample_kernel`, `_apply_sinc_resample_kernel`): reduce the two rates by their gcd,
  build `new` polyphase kernels of `2 * width + orig` taps in float64, cast to float32, correlate with stride `orig`, trim to
  `ceil(new * length / orig)` samples.  `orig == new` returns the input unchanged, like the transform.
"""
from __future__ import annotations

import math
import struct
import wave
from typing import Optional, Tuple

import numpy as np

from .prompt import PromptAudio


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, samples] in [-1, 1), sample rate)."""
    try:
        with wave.open(path, "rb") as w:
            nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
        if width == 1:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width == 2:
            x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        elif width == 3:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif width == 4:
            x = (np.frombuffer(raw, "<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM sample width {width}")
    except wave.Error:
        x, sr, nch = _read_float_wav(path)
    return np.ascontiguousarray(x.reshape(-1, nch).T), int(sr)


def _read_float_wav(path: str):
    """WAVE_FORMAT_IEEE_FLOAT (3) / EXTENSIBLE files, which the `wave` module refuses."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (other containers need a decoder this image does not have)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:      # extensible: the sub-format's first two bytes are the real tag
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt / data chunk")
    tag, nch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(payload[: len(payload) // 4 * 4], "<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(payload[: len(payload) // 8 * 8], "<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(payload[: len(payload) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} / {bits} bit")
    return x[: len(x) // nch * nch], sr, nch


def sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio `_get_sinc_resample_kernel` (sinc_interp_hann): -> (float32 [new, 2 * width + orig], width); orig / new already reduced."""
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k *= window * scale
    return k.astype(np.float32), width


def sinc_resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """torchaudio.transforms.Resample(orig_freq, new_freq)(x) on the last axis (float32)."""
    x = np.asarray(x, np.float32)
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return x
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    kern, width = sinc_resample_kernel(orig, new, lowpass_filter_width, rolloff)
    lead = x.shape[:-1]
    w = x.reshape(-1, x.shape[-1])
    length = w.shape[-1]
    w = np.pad(w, ((0, 0), (width, width + orig)))
    nfr = (w.shape[-1] - kern.shape[1]) // orig + 1
    # frames[n, f, :] = w[n, f * orig : f * orig + taps]  (a strided view; the product is one GEMM per waveform)
    st = w.strides
    frames = np.lib.stride_tricks.as_strided(w, (w.shape[0], nfr, kern.shape[1]), (st[0], st[1] * orig, st[1]), writeable=False)
    out = np.einsum("nft,pt->nfp", frames, kern, optimize=True).reshape(w.shape[0], -1).astype(np.float32)
    target = int(math.ceil(new * length / orig))
    return out[:, :target].reshape(*lead, target)


def load_and_cut_audio(path: str, max_audio_length_seconds: float, sr: Optional[int] = None) -> Tuple[np.ndarray, int]:
    """`IndexTTS2._load_and_cut_audio` (infer_v2.py:510-523): -> ([1, samples] float32, rate); rate 22 050 unless `sr` is given."""
    x, file_sr = read_wav(path)
    mono = x.mean(axis=0, dtype=np.float32) if x.shape[0] > 1 else x[0]
    target = int(sr) if sr else 22050                                    # librosa.load's default rate
    mono = sinc_resample(mono, file_sr, target)                          # (librosa: soxr_hq -- see the module docstring)
    mono = mono[: int(max_audio_length_seconds * target)]
    return mono[None, :].astype(np.float32), target


def load_prompt_audio(path: str, emotion: bool = False, max_audio_length_seconds: float = 15) -> PromptAudio:
    """A speaker prompt (infer_v2.py:628-630: 22.05 kHz for the reference mel + 16 kHz for w2v-BERT / CAMPPlus) or, with `emotion`, an
    emotion prompt (infer_v2.py:685: loaded at 16 kHz, no 22.05 kHz side)."""
    if emotion:
        a16, _ = load_and_cut_audio(path, max_audio_length_seconds, sr=16000)
        return PromptAudio(audio_16k=a16[0])
    a, sr = load_and_cut_audio(path, max_audio_length_seconds)
    return PromptAudio(audio_16k=sinc_resample(a, sr, 16000)[0], audio_22k=sinc_resample(a, sr, 22050)[0])
