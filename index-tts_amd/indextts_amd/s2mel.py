"""Host-side mirror of the reference's s2mel interface (indextts/s2mel/modules/commons.py::MyModel and
flow_matching.py::CFM), backed by libidxtts_hip.

  * S2Mel.prepare_condition(latent, codes, code_lens)  = gpt_layer + vq2emb + length_regulator
                                                          (infer_v2.py:835-849)
  * S2Mel.cfm_inference(mu, x_lens, prompt, style, f0, n_timesteps, inference_cfg_rate=..., z=...)
                                                        = models['cfm'].inference (flow_matching.py:31-55)
Constant sinusoid tables (rotary cache, timestep features) are computed here with the same torch ops
the reference uses, so they are bit-identical; everything data-dependent runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes
import math
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib
from .config import S2MelConfig


def rope_cache(seq_len: int, n_elem: int = 64, base: float = 10000.0) -> torch.Tensor:
    """precompute_freqs_cis (gpt_fast/model.py:336-345), fp32 -> [T, n_elem/2, 2]."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: (n_elem // 2)].float() / n_elem))
    t = torch.arange(seq_len)
    freqs = torch.outer(t, freqs)
    fc = torch.polar(torch.ones_like(freqs), freqs)
    return torch.stack([fc.real, fc.imag], dim=-1).to(torch.float32).contiguous()


def timestep_tables(n_timesteps: int):
    """t_span = linspace(0,1,N+1); t accumulates dt in fp32 exactly as solve_euler does (flow_matching.py:53,85-110).
    Returns (t_emb [N,256] = TimestepEmbedder.timestep_embedding(t_k), dt [N])."""
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    t = t_span[0]
    ts, dts = [], []
    for step in range(1, n_timesteps + 1):
        dt = t_span[step] - t_span[step - 1]
        ts.append(t.clone())
        dts.append(dt.clone())
        t = t + dt
    tt = torch.stack(ts)
    args = 1000 * tt[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return emb.contiguous(), torch.stack(dts).contiguous()


class S2Mel:
    def __init__(self, state_dict, cfg: S2MelConfig = S2MelConfig(), device="cuda:0", max_frames: int = 4096):
        lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP s2mel path needs a ROCm GPU device; there is no CPU fallback")
        c = _lib.S2MelConfigC()
        for name in ("hidden_dim", "num_heads", "depth", "in_channels", "content_dim", "style_dim", "wn_hidden", "wn_layers",
                     "wn_kernel", "wn_dilation_rate", "lr_channels", "lr_in_channels", "lr_num_convs", "gpt_dim",
                     "codebook_size", "codebook_dim", "codec_hidden"):
            setattr(c, name, int(getattr(cfg, name)))
        for i, v in enumerate(cfg.gpt_layer_dims):
            c.gpt_layer_dims[i] = int(v)
        c.norm_eps = float(cfg.norm_eps)
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_s2mel_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            sd = dict(state_dict)
            sd["rope_cache"] = rope_cache(min(max_frames, cfg.block_size), cfg.head_dim, cfg.rope_base)
            _lib.load_state_dict(h, sd)
        self._ws = _lib.StreamWorkspaces()

    def _workspace(self, need: int) -> torch.Tensor:
        return self._ws.get(need, self.device)       # one per stream: several acoustic stages may run at once (serving.py)

    # ------------------------------------------------------------------------------------------
    def prepare_condition(self, latent: torch.Tensor, codes: torch.Tensor, code_lens) -> tuple:
        """latent [B,M,gpt_dim] (GPT latent pass output), codes [B,M] int64, code_lens [B]
        -> (cond [B,Tg,512] with Tg = max target length, target_lengths [B]) (infer_v2.py:835-849)."""
        lib = _lib.load()
        lat = latent.to(self.device, torch.float32).contiguous()
        cd = codes.to(self.device, torch.long).contiguous()
        B, M = cd.shape
        cl = torch.as_tensor(code_lens).detach().cpu().long().reshape(-1)
        tl = (cl * 1.72).long()                                  # infer_v2.py:844
        Tg = int(tl.max())
        cond = torch.empty(B, Tg, self.cfg.lr_channels, device=self.device, dtype=torch.float32)
        cl32 = np.ascontiguousarray(cl.numpy(), dtype=np.int32)
        tl32 = np.ascontiguousarray(tl.numpy(), dtype=np.int32)
        ws = self._workspace(int(lib.idxtts_s2mel_cond_workspace_bytes(self._h, B, M, Tg)))
        _lib.check(lib.idxtts_s2mel_prepare_cond(self._h, _lib.ptr(lat), _lib.ptr(cd), cl32.ctypes.data_as(c_void_p),
                                                 tl32.ctypes.data_as(c_void_p), B, M, Tg, _lib.ptr(cond), _lib.ptr(ws), ws.numel(),
                                                 _lib.current_stream()))
        return cond, tl.to(self.device)

    def length_regulator(self, x: torch.Tensor, ylens, n_quantizers=None, f0=None) -> tuple:
        """`models['length_regulator'](S, ylens=..., n_quantizers=3, f0=None)` (length_regulator.py:90-141): x [B,M,1024], ylens [B]
        -> (cond [B, max(ylens), 512], ylens): the prompt-side call of infer_v2.py:649-652 (S_ref -> prompt_condition)."""
        if f0 is not None:
            raise NotImplementedError("f0 conditioning is disabled in IndexTTS-2 (config.yaml:76)")
        lib = _lib.load()
        xs = x.to(self.device, torch.float32).contiguous()
        B, M, _ = xs.shape
        tl = torch.as_tensor(ylens).detach().cpu().long().reshape(-1)
        Tg = int(tl.max())
        il32 = np.full(B, M, np.int32)
        tl32 = np.ascontiguousarray(tl.numpy(), dtype=np.int32)
        cond = torch.empty(B, Tg, self.cfg.lr_channels, device=self.device, dtype=torch.float32)
        ws = self._workspace(int(lib.idxtts_s2mel_cond_workspace_bytes(self._h, B, M, Tg)))
        _lib.check(lib.idxtts_s2mel_regulate(self._h, _lib.ptr(xs), il32.ctypes.data_as(c_void_p), tl32.ctypes.data_as(c_void_p), B, M, Tg,
                                             _lib.ptr(cond), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return cond, tl.to(self.device)

    def cfm_inference(self, mu: torch.Tensor, x_lens, prompt: torch.Tensor, style: torch.Tensor, f0=None, n_timesteps: int = 20,
                      temperature: float = 1.0, inference_cfg_rate: float = 0.7, z: torch.Tensor = None, prompt_lens=None):
        """mu [B,T,512], x_lens [B], prompt [B,80,Tp], style [B,192] -> mel [B,80,T] (flow_matching.py:31-55).
        `z` is the N(0,1) noise the reference draws at line 52; when omitted it is drawn here on the device."""
        if f0 is not None:
            raise NotImplementedError("f0 conditioning is disabled in IndexTTS-2 (config.yaml:76)")
        lib = _lib.load()
        mu = mu.to(self.device, torch.float32).contiguous()
        B, T, _ = mu.shape
        prompt = prompt.to(self.device, torch.float32).contiguous()
        style = style.to(self.device, torch.float32).contiguous()
        if z is None:
            z = torch.randn([B, self.cfg.in_channels, T], device=self.device) * temperature
        z = z.to(self.device, torch.float32).contiguous()
        xl = np.ascontiguousarray(torch.as_tensor(x_lens).detach().cpu().reshape(-1).numpy(), dtype=np.int32)
        if len(xl) == 1 and B > 1:
            xl = np.repeat(xl, B)
        Tp = prompt.shape[-1]
        pl = np.full(B, Tp, np.int32) if prompt_lens is None else np.ascontiguousarray(
            torch.as_tensor(prompt_lens).detach().cpu().reshape(-1).numpy(), dtype=np.int32)
        t_emb, dt = timestep_tables(n_timesteps)
        t_emb = t_emb.to(self.device)
        dt_h = np.ascontiguousarray(dt.numpy(), dtype=np.float32)
        out = torch.empty(B, self.cfg.in_channels, T, device=self.device, dtype=torch.float32)
        ws = self._workspace(int(lib.idxtts_s2mel_cfm_workspace_bytes(self._h, B, T, n_timesteps)))
        _lib.check(lib.idxtts_s2mel_cfm(self._h, _lib.ptr(mu), xl.ctypes.data_as(c_void_p), _lib.ptr(prompt), pl.ctypes.data_as(c_void_p),
                                        Tp, _lib.ptr(style), _lib.ptr(z), _lib.ptr(t_emb), dt_h.ctypes.data_as(c_void_p), n_timesteps,
                                        float(inference_cfg_rate), _lib.ptr(out), B, T, _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return out

    def estimator(self, x: torch.Tensor, prompt_x: torch.Tensor, x_lens, t: torch.Tensor, style: torch.Tensor, cond: torch.Tensor,
                  prompt_lens=None) -> torch.Tensor:
        """`cfm.estimator(x, prompt_x, x_lens, t, style, cond)` = DiT.forward (diffusion_transformer.py:186-257): x, prompt_x
        [B,80,T] (prompt_x zero beyond the prompt), x_lens [B], t [B] (one shared timestep), style [B,192], cond [B,T,512]
        -> [B,80,T].  prompt_lens: valid prompt frames per row (default: up to the last non-zero column of prompt_x)."""
        lib = _lib.load()
        x = x.to(self.device, torch.float32).contiguous()
        px = prompt_x.to(self.device, torch.float32).contiguous()
        B, C, T = x.shape
        tt = torch.as_tensor(t).detach().cpu().float().reshape(-1)
        if not bool((tt == tt[0]).all()):
            raise ValueError("the estimator is evaluated at one timestep for the whole batch")
        if prompt_lens is None:
            nz = (px != 0).any(dim=1).cpu()
            prompt_lens = [int(r.nonzero().max()) + 1 if bool(r.any()) else 0 for r in nz]
        pl = np.ascontiguousarray(torch.as_tensor(prompt_lens).detach().cpu().reshape(-1).numpy(), dtype=np.int32)
        xl = np.ascontiguousarray(torch.as_tensor(x_lens).detach().cpu().reshape(-1).numpy(), dtype=np.int32)
        half = 128       # TimestepEmbedder.timestep_embedding (diffusion_transformer.py:38-54), scale 1000
        freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
        args = 1000 * tt[:1, None] * freqs[None]
        t_emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).contiguous().to(self.device)
        mu = cond.to(self.device, torch.float32).contiguous()
        st = style.to(self.device, torch.float32).contiguous()
        out = torch.empty(B, T, C, device=self.device, dtype=torch.float32)
        ws = self._workspace(int(lib.idxtts_s2mel_cfm_workspace_bytes(self._h, B, T, 1)))
        _lib.check(lib.idxtts_s2mel_estimator(self._h, _lib.ptr(x), _lib.ptr(px), pl.ctypes.data_as(c_void_p), T, xl.ctypes.data_as(c_void_p),
                                              _lib.ptr(t_emb), _lib.ptr(st), _lib.ptr(mu), _lib.ptr(out), B, T, _lib.ptr(ws), ws.numel(),
                                              _lib.current_stream()))
        return out.transpose(1, 2)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
