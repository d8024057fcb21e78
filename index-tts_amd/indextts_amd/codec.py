"""Host-side mirror of the reference's semantic codec on the prompt path, backed by libidxtts_hip.

`SemanticCodec(state_dict).quantize(x)` is `RepCodec.quantize` (utils/maskgct/models/codec/kmeans/repcodec_model.py:179-199) as the
prompt block calls it (`_, S_ref = self.semantic_codec.quantize(spk_cond_emb)`, infer_v2.py:637): x [B,T,1024] -> (indices [B,T]
int64 -- squeezed to [T] for B = 1, as the reference does --, quantized [B,T,1024]).  `state_dict` is RepCodec's own (keys
"encoder.*", "quantizer.quantizers.0.*"; an optional "semantic_codec." prefix and weight-norm pairs are accepted and folded).
All arithmetic runs in the HIP kernels (csrc/codec.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import torch

from . import _lib
from .checkpoint import fold_weight_norm
from .config import RepCodecConfig


class SemanticCodec:
    def __init__(self, state_dict, cfg: RepCodecConfig = RepCodecConfig(), device="cuda:0"):
        lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP semantic codec needs a ROCm GPU device; there is no CPU fallback")
        sd = {(k[len("semantic_codec."):] if k.startswith("semantic_codec.") else k): torch.as_tensor(v) for k, v in state_dict.items()}
        sd = fold_weight_norm(sd)
        sd = {k: v for k, v in sd.items() if k.startswith("encoder.") or k.startswith("quantizer.quantizers.0.")}
        c = _lib.RepCodecConfigC(cfg.hidden_size, cfg.codebook_size, cfg.codebook_dim, cfg.vocos_dim, cfg.vocos_intermediate_dim,
                                 cfg.vocos_num_layers)
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_repcodec_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            _lib.load_state_dict(h, sd)
        self._ws = None

    def quantize(self, x: torch.Tensor):
        lib = _lib.load()
        x = x.to(self.device, torch.float32).contiguous()
        if x.dim() != 3 or x.shape[2] != self.cfg.hidden_size:
            raise ValueError(f"x must be [B, T, {self.cfg.hidden_size}]")
        B, T, _ = x.shape
        need = int(lib.idxtts_repcodec_workspace_bytes(self._h, B, T))
        if need == 0:
            raise RuntimeError("idxtts_repcodec_workspace_bytes returned 0")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        idx = torch.empty(B, T, dtype=torch.long, device=self.device)
        q = torch.empty(B, T, self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        _lib.check(lib.idxtts_repcodec_quantize(self._h, _lib.ptr(x), B, T, _lib.ptr(idx), _lib.ptr(q), _lib.ptr(self._ws), self._ws.numel(),
                                                _lib.current_stream()))
        # ResidualVQ returns all_indices [N=1, B, T]; quantize() squeezes dim 0 only when its size is 1 -- it always is here --
        # so the reference hands back [B, T] (repcodec_model.py:196-199)
        return idx, q

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
