"""Host-side mirror of the reference's CAMPPlus speaker encoder, backed by libidxtts_hip.

`CAMPPlus(state_dict)(feat)` is `self.campplus_model(feat.unsqueeze(0))` (infer_v2.py:251-257, 647): feat [B,T,80] -- the Kaldi fbank of
the 16 kHz prompt minus its mean over time (641-646; `indextts_amd.features.kaldi_fbank`) -- -> the style vector [B,192].
`state_dict` is the reference module's own (`campplus_cn_common.bin`).  All arithmetic runs in the HIP kernels (csrc/campplus.hip);
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import torch

from . import _lib
from .config import CamPPlusConfig


class CAMPPlus:
    def __init__(self, state_dict, cfg: CamPPlusConfig = CamPPlusConfig(), device="cuda:0"):
        lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP CAMPPlus needs a ROCm GPU device; there is no CPU fallback")
        c = _lib.CamPPlusConfigC(cfg.feat_dim, cfg.embedding_size, cfg.m_channels, cfg.growth_rate, cfg.bn_size, cfg.init_channels,
                                 len(cfg.block_layers))
        for i, (n, d) in enumerate(zip(cfg.block_layers, cfg.block_dilation)):
            c.block_layers[i] = n
            c.block_dilation[i] = d
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_campplus_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            _lib.load_state_dict(h, sd)
        self._ws = None

    def __call__(self, feat: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        x = feat.to(self.device, torch.float32).contiguous()
        if x.dim() != 3 or x.shape[2] != self.cfg.feat_dim:
            raise ValueError(f"feat must be [B, T, {self.cfg.feat_dim}]")
        B, T, _ = x.shape
        need = int(lib.idxtts_campplus_workspace_bytes(self._h, T))
        if need == 0:
            raise ValueError("CAMPPlus needs at least 8 frames")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, self.cfg.embedding_size, device=self.device, dtype=torch.float32)
        _lib.check(lib.idxtts_campplus_forward(self._h, _lib.ptr(x), B, T, _lib.ptr(out), _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
