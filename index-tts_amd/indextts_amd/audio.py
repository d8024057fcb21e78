"""Host-side mirror of the reference's `mel_fn`, backed by libidxtts_hip.

`MelSpectrogram()(audio_22k)` is `mel_spectrogram(y, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0,
fmax=None, center=False)` (s2mel/modules/audio.py:45-83 with the arguments of infer_v2.py:291-301): y [B,N] in [-1,1] on the GPU ->
log-mel [B,80,T].  The mel basis is `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` in the reference; librosa is absent here,
`slaney_mel_basis` restates its published algorithm (Slaney scale, area normalisation) and tests/test_audio_cpu.py pins it to
transformers' port of the same function.  STFT, magnitude, mel product and log run in the HIP kernels (csrc/audio.hip).
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax) (htk=False, norm='slaney') -> float32 [n_mels, n_fft/2+1]"""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, len(fftfreqs)))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class MelSpectrogram:
    def __init__(self, n_fft: int = 1024, hop_size: int = 256, win_size: int = 1024, num_mels: int = 80, sampling_rate: int = 22050,
                 fmin: float = 0.0, fmax=None, device="cuda:0"):
        lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP mel spectrogram needs a ROCm GPU device; there is no CPU fallback")
        self.num_mels = num_mels
        c = _lib.MelSpecConfigC(n_fft, hop_size, win_size, num_mels)
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_melspec_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            _lib.load_state_dict(h, {"mel_basis": torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)),
                                     "window": torch.hann_window(win_size)})
        self._ws = None

    def __call__(self, y: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        if y.device.type != "cuda":
            raise RuntimeError("y: expected a ROCm GPU tensor; the HIP path has no CPU fallback")
        y = y.to(torch.float32).contiguous()
        if y.dim() != 2:
            raise ValueError("y must be [B, samples]")
        B, N = y.shape
        T = int(lib.idxtts_melspec_frames(self._h, N))
        need = int(lib.idxtts_melspec_workspace_bytes(self._h, B, N))
        if T <= 0 or need == 0:
            raise ValueError("audio shorter than one frame")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=y.device)
        out = torch.empty(B, self.num_mels, T, device=y.device, dtype=torch.float32)
        _lib.check(lib.idxtts_melspec_forward(self._h, _lib.ptr(y), B, N, _lib.ptr(out), _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
