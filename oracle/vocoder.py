"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the BigVGAN-v2 vocoder.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
The product path (`index-tts_amd/`) never does; it fails loudly when the HIP library is missing.

Follows the reference's torch path (the one `infer_v2` runs on CPU):
  * kaiser-sinc filter          alias_free_activation/torch/filter.py:30-62
  * UpSample1d / DownSample1d   alias_free_activation/torch/resample.py:10-58, filter.py:94-101
  * SnakeBeta (log-scale)       activations.py:107-120
  * Activation1d                alias_free_activation/torch/act.py:8-30
  * AMPBlock1.forward           bigvgan.py:132-141
  * BigVGAN.forward             bigvgan.py:360-386  (use_tanh_at_final=false -> clamp, no final bias)
Pinned against the imported reference classes by tests/golden/make_golden.py -> tests/golden/vocoder_*.npz.

The arithmetic is written out index-by-index (polyphase form) rather than through
F.conv_transpose1d/F.pad so that it doubles as the specification of the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """filter.py:30-62.  Returns [kernel_size] float32 (the reference returns [1,1,K])."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if even:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    if cutoff == 0:
        return torch.zeros(kernel_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.to(torch.float32)


def aa_filter() -> torch.Tensor:
    """The one filter BigVGAN uses for both resamplers: ratio 2, 12 taps (resample.py:22-24, 46-51)."""
    return kaiser_sinc_filter1d(cutoff=0.25, half_width=0.3, kernel_size=12)


def upsample2x(x: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    """UpSample1d.forward (resample.py:28-37) for ratio=2, K=12, written as two 6-tap phases.

    replicate-pad 5/5, conv_transpose stride 2, x2, crop 15/15  ==>
      u[2j]   = 2 * sum_{q=0..5} f[2q+1] * x[clamp(j+2-q)]
      u[2j+1] = 2 * sum_{q=0..5} f[2q]   * x[clamp(j+3-q)]
    """
    B, C, T = x.shape
    xp = F.pad(x, (3, 3), mode="replicate")                 # xp[i] = x[clamp(i-3)]
    even = torch.zeros_like(x)
    odd = torch.zeros_like(x)
    for q in range(6):
        even = even + filt[2 * q + 1] * xp[..., 5 - q: 5 - q + T]   # x[j+2-q]
        odd = odd + filt[2 * q] * xp[..., 6 - q: 6 - q + T]         # x[j+3-q]
    u = torch.stack([even, odd], dim=-1).reshape(B, C, 2 * T)
    return 2.0 * u


def snake_beta(u: torch.Tensor, log_alpha: torch.Tensor, log_beta: torch.Tensor) -> torch.Tensor:
    """SnakeBeta.forward with alpha_logscale=True (activations.py:113-118)."""
    a = torch.exp(log_alpha)[None, :, None]
    b = torch.exp(log_beta)[None, :, None]
    return u + (1.0 / (b + 1e-9)) * torch.sin(u * a) ** 2


def downsample2x(v: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    """LowPassFilter1d.forward (filter.py:94-101), stride 2, replicate pad 5/6:
       out[t] = sum_k g[k] * v[clamp(2t + k - 5, 0, 2T-1)]"""
    B, C, T2 = v.shape
    T = T2 // 2
    vp = F.pad(v, (5, 6), mode="replicate")
    out = torch.zeros(B, C, T, dtype=v.dtype)
    for k in range(12):
        out = out + filt[k] * vp[..., k: k + 2 * T: 2]
    return out


def activation1d(x: torch.Tensor, log_alpha: torch.Tensor, log_beta: torch.Tensor,
                 up_filt: torch.Tensor = None, down_filt: torch.Tensor = None) -> torch.Tensor:
    """Activation1d.forward (act.py:25-30): up x2 -> SnakeBeta -> down x2.  x: [B,C,T] -> [B,C,T]."""
    up_filt = aa_filter() if up_filt is None else up_filt
    down_filt = aa_filter() if down_filt is None else down_filt
    return downsample2x(snake_beta(upsample2x(x, up_filt), log_alpha, log_beta), down_filt)


def _t(w: Dict[str, np.ndarray], key: str) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def amp_block1(w, prefix: str, x: torch.Tensor, kernel: int, dilations) -> torch.Tensor:
    """AMPBlock1.forward (bigvgan.py:132-141)."""
    for l, d in enumerate(dilations):
        xt = activation1d(x, _t(w, f"{prefix}.activations.{2*l}.act.alpha"), _t(w, f"{prefix}.activations.{2*l}.act.beta"))
        xt = F.conv1d(xt, _t(w, f"{prefix}.convs1.{l}.weight"), _t(w, f"{prefix}.convs1.{l}.bias"),
                      dilation=d, padding=(kernel * d - d) // 2)
        xt = activation1d(xt, _t(w, f"{prefix}.activations.{2*l+1}.act.alpha"), _t(w, f"{prefix}.activations.{2*l+1}.act.beta"))
        xt = F.conv1d(xt, _t(w, f"{prefix}.convs2.{l}.weight"), _t(w, f"{prefix}.convs2.{l}.bias"),
                      dilation=1, padding=(kernel - 1) // 2)
        x = xt + x
    return x


def bigvgan_forward(w, cfg, mel: torch.Tensor, clamp: bool = True) -> torch.Tensor:
    """BigVGAN.forward (bigvgan.py:360-386).  mel [B,num_mels,Tm] -> wav [B,1,Tm*256]."""
    x = F.conv1d(mel, _t(w, "conv_pre.weight"), _t(w, "conv_pre.bias"), padding=3)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.conv_transpose1d(x, _t(w, f"ups.{i}.0.weight"), _t(w, f"ups.{i}.0.bias"), stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            r = amp_block1(w, f"resblocks.{i * cfg.num_kernels + j}", x, rk, dils)
            xs = r if xs is None else xs + r
        x = xs / cfg.num_kernels
    x = activation1d(x, _t(w, "activation_post.act.alpha"), _t(w, "activation_post.act.beta"))
    x = F.conv1d(x, _t(w, "conv_post.weight"), None, padding=3)
    return torch.clamp(x, -1.0, 1.0) if clamp else x
