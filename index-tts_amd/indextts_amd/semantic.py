"""Host-side mirror of the reference's semantic feature model, backed by libidxtts_hip.

`SemanticModel(state_dict)(input_features, attention_mask)` is `IndexTTS2.get_emb` (infer_v2.py:381-408):
    vq_emb = self.semantic_model(input_features=..., attention_mask=..., output_hidden_states=True)
    feat = (vq_emb.hidden_states[17] - self.semantic_mean) / self.semantic_std
with `semantic_model = Wav2Vec2BertModel.from_pretrained("facebook/w2v-bert-2.0")` (utils/maskgct_utils.py:87-93).  `state_dict`
is that model's own `state_dict()` (only `feature_projection.*` and the first 17 `encoder.layers.*` are consumed) plus the two
vectors of wav2vec2bert_stats.pt as `semantic_mean` / `semantic_std` (= sqrt(var)), or passed as `mean=` / `std=`.
All arithmetic runs in the HIP kernels (csrc/semantic.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib
from .config import W2VBertConfig


class SemanticModel:
    def __init__(self, state_dict, cfg: W2VBertConfig = W2VBertConfig(), device="cuda:0", mean=None, std=None):
        lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP semantic model needs a ROCm GPU device; there is no CPU fallback")
        c = _lib.W2VBertConfigC(cfg.input_dim, cfg.hidden_size, cfg.num_heads, cfg.intermediate_size, cfg.num_layers, cfg.left_max,
                                cfg.right_max, cfg.conv_kernel, cfg.layer_norm_eps)
        used = tuple(f"encoder.layers.{i}." for i in range(cfg.num_layers))
        sd = {k: v for k, v in state_dict.items()
              if k.startswith("feature_projection.") or k.startswith(used) or k in ("semantic_mean", "semantic_std")}
        if mean is not None:
            sd["semantic_mean"] = torch.as_tensor(mean).reshape(-1)
        if std is not None:
            sd["semantic_std"] = torch.as_tensor(std).reshape(-1)
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_w2vbert_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            _lib.load_state_dict(h, sd)
        self._ws = None

    def __call__(self, input_features: torch.Tensor, attention_mask=None) -> torch.Tensor:
        """input_features [B,T,160] (SeamlessM4TFeatureExtractor), attention_mask [B,T] of 0/1 (right padding) or None
        -> [B,T,1024]; rows at padded frames are unspecified (the reference's values there never reach a valid frame)."""
        lib = _lib.load()
        x = input_features.to(self.device, torch.float32).contiguous()
        if x.dim() != 3 or x.shape[2] != self.cfg.input_dim:
            raise ValueError(f"input_features must be [B, T, {self.cfg.input_dim}]")
        B, T, _ = x.shape
        ln = None
        if attention_mask is not None:
            am = torch.as_tensor(attention_mask).detach().cpu().to(torch.int64)
            lens = am.sum(1)
            if not bool((am == (torch.arange(T)[None, :] < lens[:, None])).all()):
                raise ValueError("attention_mask must be right-padded (ones then zeros)")
            ln = np.ascontiguousarray(lens.numpy(), dtype=np.int32)
        need = int(lib.idxtts_w2vbert_workspace_bytes(self._h, B, T))
        if need == 0:
            raise RuntimeError("idxtts_w2vbert_workspace_bytes returned 0")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, T, self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        _lib.check(lib.idxtts_w2vbert_forward(self._h, _lib.ptr(x), c_void_p(ln.ctypes.data) if ln is not None else c_void_p(0), B, T,
                                              _lib.ptr(out), _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return out

    get_emb = __call__

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
