"""Host-side mirror of the prompt-conditioning half of the reference's `UnifiedVoice` (indextts/gpt/model_v2.py),
backed by libidxtts_hip:

  * get_conditioning(speech_conditioning_input [B,1024,T], cond_mel_lengths)        model_v2.py:627-663 -> [B,32,d]
  * get_emo_conditioning / get_emovec(emo_speech_conditioning_latent [B,T,1024], emo_cond_lengths)   665-671, 897-902 -> [B,d]
  * merge_emovec(spk_cond, emo_cond, cond_lengths, emo_cond_lengths, alpha)          904-910

Argument layouts are the reference's (get_conditioning takes the [B,1024,T] transposed tensor and transposes it back,
exactly like model_v2.py:637).  "Lengths" larger than T mean "no padding" -- `infer_v2.py:751-752` passes `shape[-1]`
of a [1,T,1024] tensor, i.e. 1024.  All arithmetic runs in the HIP kernels (csrc/cond.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib
from .config import CondModuleConfig, GPTConfig

_PREFIXES = {0: ("conditioning_encoder.", "perceiver_encoder."),
             1: ("emo_conditioning_encoder.", "emo_perceiver_encoder.", "emovec_layer.", "emo_layer.")}


class _Encoder:
    def __init__(self, state_dict, m: CondModuleConfig, dim: int, latents: int, emotion: int, model_dim: int, device):
        lib = _lib.load()
        self.device, self.emotion, self.out_shape = device, emotion, ((model_dim,) if emotion else (latents, dim))
        c = _lib.CondConfigC(m.input_size, m.output_size, m.linear_units, m.attention_heads, m.num_blocks, m.cnn_kernel,
                             dim, latents, m.perceiver_depth, m.perceiver_dim_head, m.perceiver_mult, emotion, model_dim)
        h = c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.idxtts_cond_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            sd = {k: v for k, v in state_dict.items() if k.startswith(_PREFIXES[emotion])}
            _lib.load_state_dict(h, sd)
        self._ws = None

    def __call__(self, feats: torch.Tensor, lengths) -> torch.Tensor:
        """feats [B,T,input_size] -> [B,latents,dim] (speaker) or the emotion vector [B,model_dim]."""
        lib = _lib.load()
        x = feats.to(self.device, torch.float32).contiguous()
        B, T, _ = x.shape
        ln = None
        if lengths is not None:
            ln = np.ascontiguousarray(np.broadcast_to(np.asarray(torch.as_tensor(lengths).detach().cpu()).reshape(-1), (B,)), dtype=np.int32)
        need = int(lib.idxtts_cond_workspace_bytes(self._h, B, T))
        if need == 0:
            raise RuntimeError("idxtts_cond_workspace_bytes returned 0 (a prompt needs at least 3 frames)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((B,) + self.out_shape, device=self.device, dtype=torch.float32)
        _lib.check(lib.idxtts_cond_forward(self._h, _lib.ptr(x), c_void_p(ln.ctypes.data) if ln is not None else c_void_p(0), B, T,
                                           _lib.ptr(out), _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass


class ConditioningEncoders:
    """conditioning_encoder + perceiver_encoder, emo_conditioning_encoder + emo_perceiver_encoder + emovec_layer + emo_layer
    of `UnifiedVoice.state_dict()` (model_v2.py:396-423)."""

    def __init__(self, state_dict, cfg: GPTConfig = GPTConfig(), device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP conditioning encoders need a ROCm GPU device; there is no CPU fallback")
        self.spk = _Encoder(state_dict, cfg.cond_module, cfg.model_dim, cfg.cond_latents, 0, cfg.model_dim, self.device)
        self.emo = _Encoder(state_dict, cfg.emo_cond_module, cfg.emo_perceiver_dim, 1, 1, cfg.model_dim, self.device)

    @staticmethod
    def has_weights(state_dict) -> bool:
        return any(k.startswith("conditioning_encoder.") for k in state_dict)

    def get_conditioning(self, speech_conditioning_input: torch.Tensor, cond_mel_lengths=None) -> torch.Tensor:
        return self.spk(speech_conditioning_input.transpose(1, 2), cond_mel_lengths)

    def get_emovec(self, emo_speech_conditioning_latent: torch.Tensor, emo_cond_lengths=None) -> torch.Tensor:
        return self.emo(emo_speech_conditioning_latent, emo_cond_lengths)

    def merge_emovec(self, speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths=None, emo_cond_lengths=None,
                     alpha: float = 1.0) -> torch.Tensor:
        emo_vec = self.get_emovec(emo_speech_conditioning_latent, emo_cond_lengths)
        base_vec = self.get_emovec(speech_conditioning_latent, cond_lengths)
        out = torch.empty_like(base_vec)
        _lib.check(_lib.load().idxtts_emovec_merge(_lib.ptr(out), _lib.ptr(base_vec), _lib.ptr(emo_vec), float(alpha), out.numel(),
                                                   _lib.current_stream()))
        return out
