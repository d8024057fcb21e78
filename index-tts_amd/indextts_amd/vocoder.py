"""Host-side mirror of the reference's vocoder interface, backed by libidxtts_hip.

  * `anti_alias_activation_forward(inputs, up_ftr, down_ftr, alpha, beta)` has the signature of the
    reference's pybind op `anti_alias_activation_cuda.forward`
    (alias_free_activation/cuda/anti_alias_activation.cpp:19-23; used at activation1d.py:21-26).
  * `BigVGAN(state_dict, cfg)(mel)` has the call contract of `BigVGAN.forward` (bigvgan.py:360-386)
    as used at infer_v2.py:860: mel [B,80,Tm] float32 on the GPU -> wav [B,1,Tm*256] in [-1,1].
PyTorch is only the allocator / stream provider here; all arithmetic runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import torch

from . import _lib
from .config import BigVGANConfig


def kaiser_sinc_filter12() -> torch.Tensor:
    """The 12-tap kaiser-sinc low-pass both resamplers register as a buffer (filter.py:30-62),
    computed exactly as the reference does (float32 torch.kaiser_window)."""
    import math
    kernel_size, cutoff, half_width = 12, 0.25, 0.3
    half_size = kernel_size // 2
    A = 2.285 * (half_size - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = torch.arange(-half_size, half_size) + 0.5
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).to(torch.float32)


def _require_gpu_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a ROCm GPU tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: float32 expected, got {t.dtype}")
    return t.contiguous()


def anti_alias_activation_forward(inputs: torch.Tensor, up_ftr: torch.Tensor, down_ftr: torch.Tensor,
                                  alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """Fused up(x2) -> SnakeBeta(log-scale alpha, beta) -> down(x2).  inputs [B,C,T] float32 / float16 / bfloat16 (the reference
    kernel's dtype dispatch, anti_alias_activation_cuda.cu:232-244) -> new tensor [B,C,T] of the same dtype; filters and alpha / beta
    are taken as float32 and the arithmetic is float32."""
    lib = _lib.load()
    dt = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}.get(inputs.dtype)
    if dt is None:
        raise TypeError(f"inputs: float32, float16 or bfloat16, got {inputs.dtype}")
    if dt == 0:
        x = _require_gpu_f32(inputs, "inputs")
    else:
        if inputs.device.type != "cuda":
            raise RuntimeError("inputs: expected a ROCm GPU tensor; the HIP path has no CPU fallback")
        x = inputs.contiguous()
    if x.dim() != 3:
        raise ValueError("inputs must be [B, C, T]")
    B, C, T = x.shape
    upf = _require_gpu_f32(up_ftr.reshape(-1).float(), "up_ftr")
    dnf = _require_gpu_f32(down_ftr.reshape(-1).float(), "down_ftr")
    a = _require_gpu_f32(alpha.reshape(-1).float(), "alpha")
    b = _require_gpu_f32(beta.reshape(-1).float(), "beta")
    if upf.numel() != 12 or dnf.numel() != 12 or a.numel() != C or b.numel() != C:
        raise ValueError("filters must have 12 taps and alpha/beta one value per channel")
    out = torch.empty_like(x)   # freshly allocated with the input's options, like the reference (.cu:220-223)
    _lib.check(lib.idxtts_aa_act_fwd(_lib.ptr(out), _lib.ptr(x), _lib.ptr(upf), _lib.ptr(dnf), _lib.ptr(a),
                                     _lib.ptr(b), B, C, T, dt, _lib.current_stream()))
    return out


class Conv1d:
    """Stand-alone HIP Conv1d / ConvTranspose1d (weights packed once at construction)."""

    def __init__(self, weight: torch.Tensor, bias=None, transposed_stride: int = 1):
        lib = _lib.load()
        w = weight.detach().to(torch.float32).contiguous()
        if transposed_stride > 1:
            cin, cout, k = w.shape
        else:
            cout, cin, k = w.shape
        self.cout, self.cin, self.k, self.ups = cout, cin, k, transposed_stride
        b = None if bias is None else bias.detach().to(torch.float32).contiguous()
        h = c_void_p()
        _lib.check(lib.idxtts_conv1d_create(_lib.ptr(w), _lib.ptr(b), cout, cin, k, transposed_stride, ctypes.byref(h)))
        self._h = h

    def __call__(self, x, dilation=1, pad_left=None, pad_mode=0, residual=None, scale=1.0, out=None, accumulate=False):
        lib = _lib.load()
        x = _require_gpu_f32(x, "x")
        B, cin, T = x.shape
        if cin != self.cin:
            raise ValueError(f"expected {self.cin} input channels, got {cin}")
        if pad_left is None:
            pad_left = 1 if self.ups > 1 else (self.k - 1) * dilation // 2
        if out is None:
            if accumulate:
                raise ValueError("accumulate needs an explicit out tensor")
            out = torch.empty(B, self.cout, T * self.ups, device=x.device, dtype=torch.float32)
        res = None if residual is None else _require_gpu_f32(residual, "residual")
        _lib.check(lib.idxtts_conv1d_fwd(self._h, _lib.ptr(x), _lib.ptr(out), _lib.ptr(res), B, T, dilation, pad_left,
                                         pad_mode, float(scale), int(accumulate), _lib.current_stream()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_conv1d_destroy(self._h)
        except Exception:
            pass


class BigVGAN:
    """mel -> waveform on the GPU.  `state_dict` uses the reference's keys after remove_weight_norm()."""

    def __init__(self, state_dict, cfg: BigVGANConfig = BigVGANConfig()):
        lib = _lib.load()
        self.cfg = cfg
        c = _lib.BigVGANConfigC()
        c.num_mels = cfg.num_mels
        c.upsample_initial_channel = cfg.upsample_initial_channel
        c.num_upsamples = cfg.num_upsamples
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            c.upsample_rates[i] = u
            c.upsample_kernel_sizes[i] = k
        c.num_kernels = cfg.num_kernels
        for j, (k, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            c.resblock_kernel_sizes[j] = k
            for l, d in enumerate(dils):
                c.resblock_dilations[j][l] = d
        h = c_void_p()
        _lib.check(lib.idxtts_bigvgan_create(ctypes.byref(c), ctypes.byref(h)))
        self._h = h
        sd = dict(state_dict)
        f = kaiser_sinc_filter12()
        sd.setdefault("activation_post.upsample.filter", f.reshape(1, 1, 12))
        sd.setdefault("activation_post.downsample.lowpass.filter", f.reshape(1, 1, 12))
        _lib.load_state_dict(h, sd)
        self._wss = _lib.StreamWorkspaces()

    def workspace_bytes(self, B: int, Tm: int) -> int:
        return int(_lib.load().idxtts_bigvgan_workspace_bytes(self._h, B, Tm))

    def forward(self, mel: torch.Tensor, clamp: bool = True, stage: int = 0, lengths=None):
        """mel [B,num_mels,Tm] -> wav [B,1,Tm*256].  `stage` (1..6) additionally returns that stage's output.
        lengths (optional, [B] ints <= Tm): a ragged batch -- row b equals the call on mel[b:b+1, :, :lengths[b]] alone
        (valid samples: the first lengths[b] * 256); the mel is zeroed beyond each row's length here."""
        lib = _lib.load()
        mel = _require_gpu_f32(mel, "mel")
        if mel.dim() != 3 or mel.shape[1] != self.cfg.num_mels:
            raise ValueError(f"mel must be [B, {self.cfg.num_mels}, Tm]")
        B, _, Tm = mel.shape
        wav = torch.empty(B, 1, Tm * self.cfg.total_upsample, device=mel.device, dtype=torch.float32)
        if B == 0 or Tm == 0:
            return wav
        ws = self._wss.get(self.workspace_bytes(B, Tm), mel.device)     # one per stream
        if lengths is not None:
            if stage:
                raise ValueError("stage output is not available for ragged batches")
            lens = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1)
            if lens.numel() != B or int(lens.min()) < 0 or int(lens.max()) > Tm:
                raise ValueError("lengths must be [B] with 0 <= lengths[b] <= Tm")
            lens_d = lens.to(mel.device)
            mel = (mel * (torch.arange(Tm, device=mel.device)[None, None, :] < lens_d[:, None, None])).contiguous()
            _lib.check(lib.idxtts_bigvgan_fwd_ragged(self._h, _lib.ptr(mel), _lib.ptr(lens_d), _lib.ptr(wav), B, Tm, _lib.ptr(ws),
                                                     ws.numel(), int(clamp), _lib.current_stream()))
            return wav
        stage_out = None
        if stage:
            T = Tm
            for u in self.cfg.upsample_rates[:stage]:
                T *= u
            stage_out = torch.empty(B, self.cfg.channels(stage), T, device=mel.device, dtype=torch.float32)
        _lib.check(lib.idxtts_bigvgan_fwd(self._h, _lib.ptr(mel), _lib.ptr(wav), B, Tm, _lib.ptr(ws),
                                          ws.numel(), int(clamp), int(stage), _lib.ptr(stage_out),
                                          _lib.current_stream()))
        return (wav, stage_out) if stage else wav

    __call__ = forward

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
