"""Synthetic state dicts in the reference's key layout (SURVEY.md §8a "Weight/state-dict layouts").

Every function returns `dict[str, np.ndarray(float32)]` whose keys and shapes are exactly those of
the reference module's `state_dict()` on the inference path, so the same dict can be
 - loaded into the reference classes (`tests/golden/make_golden.py`, in the build container only),
 - fed to the CPU oracle (`oracle/`), and
 - handed tensor-by-tensor to the C-ABI (`idxtts_ctx_load_tensor`), which is also how a real
   checkpoint (`bigvgan_generator.pt['generator']`, `gpt.pth['model']`, `s2mel.pth['net']`) would go in.
Weight-norm parametrised layers are stored folded (w = g*v/||v||), i.e. after
`remove_weight_norm()` (infer_v2.py:263) for BigVGAN and after export-time folding for s2mel.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import synth
from .config import BigVGANConfig, GPTConfig, S2MelConfig

Weights = Dict[str, np.ndarray]


# --------------------------------------------------------------------------------------
# BigVGAN (bigvgan.py:266-386)
# --------------------------------------------------------------------------------------
def synth_bigvgan_weights(cfg: BigVGANConfig, tag: str = "bigvgan") -> Weights:
    w: Weights = {}

    def conv(name, cout, cin, k, gain, bias=True):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (cout, cin, k), cin * k, gain)
        if bias:
            w[f"{name}.bias"] = synth.uniform(f"{tag}/{name}.bias", (cout,), 0.05)

    c0 = cfg.upsample_initial_channel
    conv("conv_pre", c0, cfg.num_mels, 7, gain=0.35)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = cfg.channels(i), cfg.channels(i + 1)
        # ConvTranspose1d weight is [Cin, Cout, k]; each output sample sees k/u taps of every input channel
        w[f"ups.{i}.0.weight"] = synth.fan_in_uniform(f"{tag}/ups.{i}.0.weight", (cin, cout, k), cin * k // u, 0.9)
        w[f"ups.{i}.0.bias"] = synth.uniform(f"{tag}/ups.{i}.0.bias", (cout,), 0.05)
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            r = i * cfg.num_kernels + j
            for l in range(len(dils)):
                conv(f"resblocks.{r}.convs1.{l}", cout, cout, rk, gain=0.6)
                conv(f"resblocks.{r}.convs2.{l}", cout, cout, rk, gain=0.45)
            for a in range(2 * len(dils)):
                w[f"resblocks.{r}.activations.{a}.act.alpha"] = synth.uniform(
                    f"{tag}/resblocks.{r}.activations.{a}.act.alpha", (cout,), 0.6)
                w[f"resblocks.{r}.activations.{a}.act.beta"] = synth.uniform(
                    f"{tag}/resblocks.{r}.activations.{a}.act.beta", (cout,), 0.6, offset=0.4)
    cl = cfg.channels(cfg.num_upsamples)
    w["activation_post.act.alpha"] = synth.uniform(f"{tag}/activation_post.act.alpha", (cl,), 0.6)
    w["activation_post.act.beta"] = synth.uniform(f"{tag}/activation_post.act.beta", (cl,), 0.6, offset=0.4)
    conv("conv_post", 1, cl, 7, gain=0.25, bias=False)
    return w


def synth_mel(name: str, batch: int, num_mels: int, frames: int) -> np.ndarray:
    """Log-mel-like input: BASELINE.md config 2 uses randn*1.5-4; here the same range, uniform."""
    return synth.uniform(name, (batch, num_mels, frames), scale=2.6, offset=-4.0)
