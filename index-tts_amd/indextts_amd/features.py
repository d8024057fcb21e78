"""Host-side audio features of the reference's prompt block (infer_v2.py:630-646) -- numpy on the CPU, as in the reference.

  * `seamless_m4t_features(waveforms)`   = `SeamlessM4TFeatureExtractor.from_pretrained("facebook/w2v-bert-2.0")(audio_16k,
        sampling_rate=16000, return_tensors="pt")` (infer_v2.py:633, 680): Kaldi fbank (80 bins, 25 ms povey window, 10 ms shift,
        pre-emphasis 0.97, samples scaled to 16-bit range), per-utterance mean / variance normalisation of every mel bin, frames padded
        to a multiple of 2 and stacked in pairs -> input_features [B, T/2, 160], attention_mask [B, T/2].
  * `kaldi_fbank(waveform, ...)`         = `torchaudio.compliance.kaldi.fbank(audio_16k, num_mel_bins=80, dither=0,
        sample_frequency=16000)` (infer_v2.py:641-644; CAMPPlus input before its mean subtraction): the same algorithm on the
        samples as they are (`scale=1.0`).
The extractor class is a third-party dependency of the reference (transformers); this restates its published algorithm
(feature_extraction_seamless_m4t.py, audio_utils.spectrogram / mel_filter_bank with mel_scale="kaldi") vectorised over frames, and
tests/test_features_cpu.py pins it to the container's own SeamlessM4TFeatureExtractor.
"""
from __future__ import annotations

import functools
from typing import Dict, Sequence

import numpy as np


try:
    from threadpoolctl import threadpool_limits as _threadpool_limits

    def _one_blas_thread():
        return _threadpool_limits(limits=1, user_api="blas")
except ImportError:      # pragma: no cover -- threadpoolctl is optional
    import contextlib

    def _one_blas_thread():
        return contextlib.nullcontext()


def _kaldi_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def kaldi_mel_filters(num_bins: int = 80, fft_length: int = 512, sampling_rate: int = 16000, low_freq: float = 20.0,
                      high_freq: float = 0.0) -> np.ndarray:
    """[fft_length/2 + 1, num_bins] triangular filters, triangular in MEL space (Kaldi's get_mel_banks)."""
    high = high_freq if high_freq > 0 else sampling_rate / 2 + high_freq
    mel_pts = np.linspace(_kaldi_mel(low_freq), _kaldi_mel(high), num_bins + 2)
    nfb = fft_length // 2 + 1
    fft_mel = _kaldi_mel(sampling_rate / fft_length * np.arange(nfb))
    diff = np.diff(mel_pts)
    slopes = mel_pts[None, :] - fft_mel[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def povey_window(n: int = 400) -> np.ndarray:
    return np.power(np.hanning(n), 0.85)


@functools.lru_cache(maxsize=8)
def _fbank_tables(num_mel_bins: int, frame_length: int, fft_length: int, sampling_rate: int, low_freq: float):
    """(window [frame_length], mel filters TRANSPOSED [num_bins, fft_length/2 + 1], C-contiguous) -- built once per geometry."""
    return povey_window(frame_length), np.ascontiguousarray(kaldi_mel_filters(num_mel_bins, fft_length, sampling_rate, low_freq).T)


def kaldi_fbank(waveform: np.ndarray, sampling_rate: int = 16000, num_mel_bins: int = 80, frame_length: int = 400, hop_length: int = 160,
                fft_length: int = 512, preemphasis: float = 0.97, low_freq: float = 20.0, scale: float = 1.0,
                mel_floor: float = 1.192092955078125e-07) -> np.ndarray:
    """Log mel filter-bank energies [frames, num_mel_bins] (snip_edges: frames that fit entirely; no dither).
    (Host time matters: a request's prompt block waits for three of these.  Frames are a strided view, every per-frame step is in
    place, the window / filter tables are cached, and the 1500 x 257 x 80 filter product runs on ONE BLAS thread.)"""
    x = np.asarray(waveform, dtype=np.float32)
    if x.ndim == 2:
        x = x[0]                                     # left channel, as the extractor does
    x = np.squeeze(x).astype(np.float64) * scale
    if x.size < frame_length:
        return np.zeros((0, num_mel_bins), np.float32)
    n = 1 + (x.size - frame_length) // hop_length
    window, filters_t = _fbank_tables(num_mel_bins, frame_length, fft_length, sampling_rate, float(low_freq))
    fr = np.lib.stride_tricks.as_strided(x, shape=(n, frame_length), strides=(hop_length * x.strides[0], x.strides[0]), writeable=False)
    fr = fr - fr.mean(axis=1, keepdims=True)         # remove_dc_offset (a fresh array: the view is read-only)
    pre = np.empty_like(fr)
    np.multiply(fr[:, :-1], preemphasis, out=pre[:, 1:])
    np.subtract(fr[:, 1:], pre[:, 1:], out=pre[:, 1:])
    pre[:, 0] = fr[:, 0] * (1.0 - preemphasis)
    pre *= window[None, :]
    spec = np.fft.rfft(pre, n=fft_length, axis=1).astype(np.complex64)      # the extractor stores complex64 before |.|^2
    power = np.abs(spec).astype(np.float64)
    np.square(power, out=power)
    with _one_blas_thread():      # 31 MFLOP: a threaded BLAS call costs more in waking (and, in a CPU-limited container, in
        mel = np.dot(power, filters_t.T)      # oversubscribing) its 64-thread pool than the product takes on one core
    np.maximum(mel, mel_floor, out=mel)
    return np.log(mel, out=mel).astype(np.float32)


def seamless_m4t_features(waveforms: Sequence[np.ndarray], sampling_rate: int = 16000, stride: int = 2, padding_value: float = 0.0) -> Dict[str, np.ndarray]:
    """waveforms: one 1-D array or a list of them (16 kHz, [-1, 1]).  Returns float32 input_features [B, T', 80 * stride] and
    int32 attention_mask [B, T'] (right padding), like the extractor with its defaults (padding=True, pad_to_multiple_of=2)."""
    if sampling_rate != 16000:
        raise ValueError("the w2v-bert-2.0 extractor is defined for 16 kHz audio")
    if isinstance(waveforms, np.ndarray) and waveforms.ndim == 1:
        waveforms = [waveforms]
    feats = []
    for wav in waveforms:
        f = kaldi_fbank(np.asarray(wav, np.float32), scale=float(2 ** 15))
        if f.shape[0] < 2:
            raise ValueError("the audio is shorter than two 25 ms frames")
        f = (f - f.mean(0, keepdims=True)) / np.sqrt(f.var(0, ddof=1, keepdims=True) + 1e-7)        # per mel bin, ddof = 1
        feats.append(f.astype(np.float32))
    tmax = max(f.shape[0] for f in feats)
    tmax += (-tmax) % 2                              # pad_to_multiple_of = 2
    B, nb = len(feats), feats[0].shape[1]
    x = np.full((B, tmax, nb), padding_value, np.float32)
    m = np.zeros((B, tmax), np.int32)
    for b, f in enumerate(feats):
        x[b, : f.shape[0]] = f
        m[b, : f.shape[0]] = 1
    tmax -= tmax % stride
    x, m = x[:, :tmax], m[:, :tmax]
    return {"input_features": x.reshape(B, tmax // stride, nb * stride), "attention_mask": m[:, np.arange(tmax) % stride == 1]}
