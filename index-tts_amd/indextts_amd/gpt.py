"""Host-side mirror of the reference's GPT interface (indextts/gpt/model_v2.py), backed by libidxtts_hip.

`UnifiedVoice` keeps the reference's method names and argument meaning for the hot path:
  * prepare_gpt_inputs(conditional_latents, text_inputs)           model_v2.py:725-794
  * inference_speech(..., text_inputs, emo_vec=..., **generate kw)  model_v2.py:796-895 (greedy path)
  * forward(...) -> latent                                          model_v2.py:673-723
  * accel_engine-style generate(input_ids, max_new_tokens, ..., attention_mask, tts_embeddings, ...)
                                                                    accel/accel_engine.py:378-645 (plugin slot)
  * get_conditioning / get_emovec / merge_emovec                      model_v2.py:627-671, 897-910 (cond.py, csrc/cond.hip)
The conformer / perceiver conditioning encoders are built when the state dict carries their weights; `inference_speech`
then accepts the prompt features [B,T,1024] like the reference does, or -- hoisted out of the segment loop, where the
reference recomputes them per segment (infer_v2.py:748-765) -- the ready `speech_conditioning_latent` [B,32,d] + `emo_vec`.
All arithmetic runs in the HIP kernels; numpy/torch only build int32 index arrays and own the buffers.
"""
from __future__ import annotations

import ctypes
import warnings
from ctypes import c_void_p
from typing import Optional

import numpy as np
import torch

from . import _lib
from .config import GPTConfig


def _i32(dev, arr) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(dev)


WEIGHT_FORMATS = {"f32": 0, "bf16": 1, "fp8": 2}
KV_FORMATS = {"f32": 0, "bf16": 1}


class UnifiedVoice:
    MAX_WORKSPACES = 4

    def __init__(self, state_dict, cfg: GPTConfig = GPTConfig(), device="cuda:0", weight_format: str = "f32", keep_effective: bool = False,
                 kv_format: str = None):
        """weight_format: storage of the GPT's linear weights -- "f32" (default), "bf16", or "fp8" (e4m3 + power-of-two scale
        per output channel): the reference's `use_fp16` switch (infer_v2.py:145-146) / BASELINE configs[4].  The weights are
        rounded ONCE at load (`idxtts_gpt_quantize_weights`); the arithmetic stays fp32, and every pass runs that one rounded
        model.  keep_effective: keep `self.effective_state_dict` (numpy, reference keys) = exactly that model, for parity checks.
        kv_format: storage of the KV cache of the cached generation, "f32" or "bf16" (`idxtts_gpt_set_kv_format`: keys / values
        rounded to bf16 once, when produced; fp32 arithmetic).  Default: "f32" with fp32 weights, "bf16" with compact weights --
        the reference's `use_fp16` halves weights and cache together."""
        lib = _lib.load()
        if weight_format not in WEIGHT_FORMATS:
            raise ValueError(f"weight_format must be one of {sorted(WEIGHT_FORMATS)}")
        self.weight_format = weight_format
        if kv_format is None:
            kv_format = "f32" if weight_format == "f32" else "bf16"
        if kv_format not in KV_FORMATS:
            raise ValueError(f"kv_format must be one of {sorted(KV_FORMATS)}")
        self.kv_format = kv_format
        self.effective_state_dict = None
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP GPT path needs a ROCm GPU device; there is no CPU fallback")
        c = _lib.GPTConfigC(cfg.model_dim, cfg.heads, cfg.layers, cfg.number_mel_codes, cfg.number_text_tokens,
                            cfg.start_mel_token, cfg.stop_mel_token, cfg.mel_pos_len, cfg.text_pos_len)
        h = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.idxtts_gpt_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            sd = {k: v for k, v in state_dict.items() if self._on_path(k)}

            def hook(ctx):
                if WEIGHT_FORMATS[weight_format]:
                    _lib.check(lib.idxtts_gpt_quantize_weights(ctx, WEIGHT_FORMATS[weight_format]))
                if keep_effective:
                    self.effective_state_dict = {k: _lib.get_tensor(ctx, k, tuple(v.shape)) for k, v in sd.items()}
            _lib.load_state_dict(h, sd, before_finalize=hook)
            _lib.check(lib.idxtts_gpt_set_kv_format(h, KV_FORMATS[kv_format]))
        from .cond import ConditioningEncoders
        self.cond = ConditioningEncoders(state_dict, cfg, device=self.device) if ConditioningEncoders.has_weights(state_dict) else None
        se = state_dict["speed_emb.weight"]
        self.speed_emb = (se if isinstance(se, torch.Tensor) else torch.from_numpy(np.asarray(se))).float().to(self.device)
        import threading
        self._ws = None
        self._ws_lock = threading.Lock()
        self._conds_cache = {}
        self._conds_lock = threading.Lock()
        self.stop_mel_token = cfg.stop_mel_token
        self.start_mel_token = cfg.start_mel_token
        self.accel_engine = self       # the reference selects `self.accel_engine.generate` (model_v2.py:871)

    def set_kv_format(self, kv_format: str) -> None:
        """Switch the KV-cache storage between generations ("f32" | "bf16")."""
        if kv_format not in KV_FORMATS:
            raise ValueError(f"kv_format must be one of {sorted(KV_FORMATS)}")
        _lib.check(_lib.load().idxtts_gpt_set_kv_format(self._h, KV_FORMATS[kv_format]))
        self.kv_format = kv_format

    @staticmethod
    def _on_path(key: str) -> bool:
        return key.startswith(("gpt.h.", "gpt.ln_f.", "final_norm.", "mel_head.", "mel_embedding.", "text_embedding.",
                               "mel_pos_embedding.", "text_pos_embedding.", "speed_emb."))

    # ------------------------------------------------------------------------------------------
    def _workspace(self, B: int, S: int, max_new: int) -> torch.Tensor:
        need = int(_lib.load().idxtts_gpt_workspace_bytes(self._h, B, S, max_new))
        if need == 0:
            raise RuntimeError("idxtts_gpt_workspace_bytes returned 0")
        return self._workspace_bytes(need)

    def _workspace_bytes(self, need: int) -> torch.Tensor:
        # one workspace per HIP stream (calls on one stream are ordered, so they can share it; generate() may run concurrently
        # on several streams from several host threads); least-recently-used streams beyond MAX_WORKSPACES are dropped, so
        # short-lived threads / streams cannot pile up KV arenas
        import collections
        if self._ws is None:
            self._ws = collections.OrderedDict()
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        with self._ws_lock:
            ws = self._ws.pop(key, None)
            if ws is None or ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
            while len(self._ws) > self.MAX_WORKSPACES:
                self._ws.popitem(last=False)
        return ws

    def _embed(self, rows: int, text_ids=None, text_pos=None, mel_ids=None, mel_pos=None, extra=None, extra_idx=None):
        # nn.Embedding raises IndexError on an out-of-range id (the reference's behaviour); the gather kernel would read past
        # its table instead, so the same check happens here, on the host index arrays, before anything is launched
        cfg = self.cfg
        for name, arr, hi in (("text token id", text_ids, cfg.number_text_tokens + 1), ("text position", text_pos, cfg.text_pos_len),
                              ("mel code", mel_ids, cfg.number_mel_codes), ("mel position", mel_pos, cfg.mel_pos_len),
                              ("conditioning row", extra_idx, 0 if extra is None else extra.shape[0])):
            if arr is not None and len(arr) and int(np.max(arr)) >= hi:
                raise IndexError(f"{name} {int(np.max(arr))} is out of range (table has {hi} rows)")
        out = torch.empty(rows, self.cfg.model_dim, device=self.device, dtype=torch.float32)
        idx = [None if a is None else _i32(self.device, a) for a in (text_ids, text_pos, mel_ids, mel_pos, extra_idx)]
        ex = None if extra is None else extra.to(self.device, torch.float32).contiguous()
        _lib.check(_lib.load().idxtts_gpt_embed(self._h, _lib.ptr(out), rows, _lib.ptr(idx[0]), _lib.ptr(idx[1]), _lib.ptr(idx[2]),
                                                _lib.ptr(idx[3]), _lib.ptr(ex), _lib.ptr(idx[4]), _lib.current_stream()))
        return out

    # ---- prompt conditioning (model_v2.py:627-671, 897-910): reference names and argument layouts ------------------
    def _need_cond(self):
        if self.cond is None:
            raise RuntimeError("this UnifiedVoice was built without conditioning_encoder / perceiver_encoder weights")
        return self.cond

    def get_conditioning(self, speech_conditioning_input: torch.Tensor, cond_mel_lengths=None) -> torch.Tensor:
        """[B,1024,T] (transposed features, as model_v2.py:819 passes them) -> [B,32,d]."""
        return self._need_cond().get_conditioning(speech_conditioning_input, cond_mel_lengths)

    def get_emovec(self, emo_speech_conditioning_latent: torch.Tensor, emo_cond_lengths=None) -> torch.Tensor:
        return self._need_cond().get_emovec(emo_speech_conditioning_latent, emo_cond_lengths)

    def merge_emovec(self, speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths=None, emo_cond_lengths=None, alpha=1.0):
        return self._need_cond().merge_emovec(speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths, emo_cond_lengths, alpha)

    def conds_latent(self, speech_conditioning_latent: torch.Tensor, emo_vec: torch.Tensor) -> torch.Tensor:
        """cat(latent + emo_vec, speed_emb(1), speed_emb(0)) -> [B, 34, d]   (model_v2.py:830-834)."""
        lat = speech_conditioning_latent.to(self.device, torch.float32)
        ev = emo_vec.to(self.device, torch.float32)
        B = lat.shape[0]
        # One prompt is decoded many times (segments, batches, serving lanes): the sum is kept per (latent, emotion vector) storage so that
        # no torch elementwise kernel runs on a decode lane's stream per call (torch's own fp32 kernels beside bf16 MFMAs are
        # outside the NOPK build switch); a consumer on another stream waits for the event of the stream that produced the entry.
        # (keyed on storage address + torch's version counter: a write that bypasses the counter -- `.data.copy_`, a kernel or collective
        #  writing through the raw pointer -- is not seen; prompt conditioning tensors are built once and not written again)
        key = (lat.data_ptr(), lat._version, tuple(lat.shape), tuple(lat.stride()), ev.data_ptr(), ev._version, tuple(ev.shape), tuple(ev.stride()))
        cur = torch.cuda.current_stream(self.device)
        with self._conds_lock:      # decode lanes call this from several host threads
            hit = self._conds_cache.get(key)
        if hit is not None:
            cur.wait_event(hit[1])
            hit[0].record_stream(cur)      # allocated on the producer's stream, read on this one: not reused while a lane still reads it
            return hit[0]
        se = self.speed_emb
        conds = torch.cat([lat + ev[:, None, :], se[1].expand(B, 1, -1), se[0].expand(B, 1, -1)], dim=1).contiguous()
        done = torch.cuda.Event()
        done.record(cur)
        with self._conds_lock:
            if len(self._conds_cache) >= 16:
                self._conds_cache.pop(next(iter(self._conds_cache)), None)
            self._conds_cache[key] = (conds, done, lat.untyped_storage(), ev.untyped_storage())      # the storages are kept alive: a data_ptr is not reused
        return conds

    def prepare_gpt_inputs(self, conditional_latents: torch.Tensor, text_inputs: torch.Tensor):
        """model_v2.py:725-794 -> (fake_inputs [B,P+1], inputs_embeds [B,P,d] on the GPU, attention_mask [B,P+1])."""
        cfg = self.cfg
        text = text_inputs.detach().cpu().numpy().astype(np.int64)
        B, L = text.shape
        nc = conditional_latents.shape[1]
        single = conditional_latents.shape[0] == 1
        if not single and conditional_latents.shape[0] != B:
            raise AssertionError(f"batch size mismatch: {conditional_latents.shape[0]} vs {B}")
        P = nc + L + 2
        tid = -np.ones((B, P), np.int32)
        tpos = -np.ones((B, P), np.int32)
        eidx = -np.ones((B, P), np.int32)
        mask = np.ones((B, P + 1), np.int64)
        for i in range(B):
            ti = text[i]
            ti = ti[(ti != cfg.stop_text_token) & (ti != cfg.start_text_token)]
            ti = np.concatenate([[cfg.start_text_token], ti, [cfg.stop_text_token]])
            pad = L + 2 - len(ti)
            mask[i, :pad] = 0
            eidx[i, pad:pad + nc] = np.arange(nc) + (0 if single else i * nc)
            tid[i, pad + nc:] = ti
            tpos[i, pad + nc:] = np.arange(len(ti))
        extra = conditional_latents.reshape(-1, cfg.model_dim)
        emb = self._embed(B * P, text_ids=tid.reshape(-1), text_pos=tpos.reshape(-1), extra=extra, extra_idx=eidx.reshape(-1))
        fake = torch.ones(B, P + 1, dtype=torch.long)
        fake[:, -1] = cfg.start_mel_token
        return fake, emb.view(B, P, cfg.model_dim), torch.from_numpy(mask)

    # ------------------------------------------------------------------------------------------
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int = 100, temperature: float = 1.0, top_k: int = 50,
                 top_p: float = 1.0, stop_tokens=None, attention_mask: Optional[torch.Tensor] = None,
                 tts_embeddings: Optional[torch.Tensor] = None, tts_mel_embedding=None, tts_text_pos_embedding=None,
                 repetition_penalty: float = 10.0, return_logits: bool = False, use_graph: bool = True,
                 do_sample: bool = False, sampler: str = "hf", exp_noise: Optional[torch.Tensor] = None,
                 generator: Optional[torch.Generator] = None, forced_codes: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The accel-engine plugin contract (accel_engine.py:378-645): returns LongTensor [B, P+1+generated]
        (prompt ids followed by the generated codes, padded with the stop token).

        do_sample=False: greedy (temperature / top_k / top_p ignored).  do_sample=True: multinomial sampling,
        sampler="hf" = HF `_sample` with its warpers (repetition penalty -> temperature -> top-k -> top-p -> multinomial,
        transformers_generation_utils.py:1036-1044, 3222-3250), sampler="accel" = the accel engine's own Sampler
        (softmax(logits / T) / Exp(1) noise, argmax; accel_engine.py:648-659).  torch.multinomial(probs, 1) is
        argmax(probs / q) with q ~ Exp(1): `exp_noise` [max_new_tokens, B, V] supplies the draws, or `generator` draws them on the
        CPU with one exponential_() per step, the order HF consumes them; with neither the kernels generate them (see _noise).
        forced_codes [B, max_new_tokens] (greedy only; a parity instrument, `idxtts_gpt_generate_forced`): the sequence is continued with
        these tokens while the returned codes are each step's own argmax -- with return_logits, the logits of a GIVEN token sequence."""
        if tts_embeddings is None:
            raise ValueError("tts_embeddings ([pad][cond][text] prompt embeddings) is required")
        if stop_tokens is not None and list(stop_tokens) != [self.cfg.stop_mel_token]:
            raise ValueError("stop_tokens must be [stop_mel_token]")
        emb = tts_embeddings.to(self.device, torch.float32).contiguous()
        B, P, d = emb.shape
        if input_ids.shape != (B, P + 1):
            raise ValueError("input_ids must be [B, P+1] (fake prefix + start_mel_token)")
        pad_left = np.zeros(B, np.int32)
        if attention_mask is not None:
            am = attention_mask.detach().cpu().numpy()
            pad_left = (am[:, :P] == 0).sum(1).astype(np.int32)
        codes = torch.full((B, max_new_tokens), self.cfg.stop_mel_token, dtype=torch.long, device=self.device)
        V = self.cfg.number_mel_codes
        logits = torch.zeros(max_new_tokens, B, V, device=self.device) if return_logits else None
        ws = self._workspace(B, P + 1, max_new_tokens)
        n = ctypes.c_int(0)
        if do_sample and forced_codes is not None:
            raise ValueError("forced_codes is a greedy-mode instrument")
        if do_sample:
            if sampler not in ("hf", "accel"):
                raise ValueError("sampler must be 'hf' or 'accel'")
            noise, seed = self._noise(exp_noise, generator, (max_new_tokens, B, V))
            sc = _lib.SamplingC(mode=1 if sampler == "hf" else 2, temperature=float(temperature),
                                top_k=int(top_k or 0) if sampler == "hf" else 0,
                                top_p=float(top_p if top_p is not None else 1.0) if sampler == "hf" else 1.0,
                                exp_noise=noise.data_ptr() if noise is not None else None, seed=seed)
            _lib.check(_lib.load().idxtts_gpt_generate_sampled(
                self._h, _lib.ptr(emb), pad_left.ctypes.data_as(c_void_p), B, P, max_new_tokens, float(repetition_penalty),
                ctypes.byref(sc), _lib.ptr(codes), ctypes.byref(n), _lib.ptr(logits), _lib.ptr(ws), ws.numel(), int(use_graph),
                _lib.current_stream()))
        elif forced_codes is not None:
            fc = forced_codes.to(self.device, torch.long).contiguous()
            if tuple(fc.shape) != (B, max_new_tokens):
                raise ValueError(f"forced_codes must be {(B, max_new_tokens)}")
            _lib.check(_lib.load().idxtts_gpt_generate_forced(
                self._h, _lib.ptr(emb), pad_left.ctypes.data_as(c_void_p), B, P, max_new_tokens, float(repetition_penalty), _lib.ptr(fc),
                _lib.ptr(codes), ctypes.byref(n), _lib.ptr(logits), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        else:
            _lib.check(_lib.load().idxtts_gpt_generate(
                self._h, _lib.ptr(emb), pad_left.ctypes.data_as(c_void_p), B, P, max_new_tokens, float(repetition_penalty),
                _lib.ptr(codes), ctypes.byref(n), _lib.ptr(logits), _lib.ptr(ws), ws.numel(), int(use_graph), _lib.current_stream()))
        out = torch.cat([input_ids.to(self.device), codes[:, : n.value]], dim=1)
        if return_logits:
            return out, logits[: n.value].permute(1, 0, 2).contiguous()
        return out

    def _noise(self, exp_noise, generator, shape):
        """The Exp(1) draws of a sampled generation: (device tensor | None, seed).  An explicit `exp_noise`, or a torch `generator`
        (one exponential_() per step, the order HF consumes them: reproduces the reference's stream under that seed); with neither,
        the kernels draw from a counter-based generator keyed by a 63-bit seed taken from torch's global RNG (so torch.manual_seed
        still makes a run repeatable) -- no [steps, B, V] tensor is materialised."""
        if exp_noise is None and generator is None:
            return None, int(torch.randint(0, 2 ** 62, (1,)).item())
        if exp_noise is None:
            exp_noise = torch.stack([torch.empty(shape[1:]).exponential_(1, generator=generator) for _ in range(shape[0])])
        if tuple(exp_noise.shape) != tuple(shape):
            raise ValueError(f"exp_noise must be {tuple(shape)}")
        return exp_noise.to(self.device, torch.float32).contiguous(), 0

    def generate_beam(self, input_ids: torch.Tensor, max_new_tokens: int, attention_mask: Optional[torch.Tensor], tts_embeddings: torch.Tensor,
                      num_beams: int = 3, do_sample: bool = True, temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0,
                      repetition_penalty: float = 1.0, length_penalty: float = 1.0, early_stopping: bool = False,
                      exp_noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, use_graph: bool = True) -> torch.Tensor:
        """HF `generate(num_beams > 1)` of the reference (model_v2.py:885-889 -> transformers_generation_utils.py:3325-3516):
        beam search, or beam-sample when do_sample.  Returns LongTensor [B, P+1+n]: the best hypothesis per utterance.
        Sampling draws 2 * num_beams candidates per utterance without replacement: torch.multinomial == top-k of probs / q with
        q ~ Exp(1) from one exponential_() per step on a [B, num_beams * V] tensor -- `exp_noise` [max_new_tokens, B, num_beams*V]
        supplies them, or `generator` draws them on the CPU in that order; with neither the kernels generate them (see _noise)."""
        emb = tts_embeddings.to(self.device, torch.float32).contiguous()
        B, P, d = emb.shape
        if input_ids.shape != (B, P + 1):
            raise ValueError("input_ids must be [B, P+1] (fake prefix + start_mel_token)")
        if not (2 <= int(num_beams) <= 8) or B * int(num_beams) > 64:
            raise ValueError("2 <= num_beams <= 8 and B * num_beams <= 64")
        if early_stopping not in (False, True):
            raise NotImplementedError('early_stopping="never" is not implemented')
        pad_left = np.zeros(B, np.int32)
        if attention_mask is not None:
            pad_left = (attention_mask.detach().cpu().numpy()[:, :P] == 0).sum(1).astype(np.int32)
        V, nb = self.cfg.number_mel_codes, int(num_beams)
        noise, seed = self._noise(exp_noise, generator, (max_new_tokens, B, nb * V)) if do_sample else (None, 0)
        lib = _lib.load()
        need = int(lib.idxtts_gpt_beam_workspace_bytes(self._h, B, nb, P + 1, max_new_tokens))
        if need == 0:
            raise RuntimeError("idxtts_gpt_beam_workspace_bytes returned 0")
        ws = self._workspace_bytes(need)
        codes = torch.full((B, max_new_tokens), self.cfg.stop_mel_token, dtype=torch.long, device=self.device)
        bc = _lib.BeamC(num_beams=nb, do_sample=int(bool(do_sample)), temperature=float(temperature), top_k=int(top_k or 0),
                        top_p=float(top_p if top_p is not None else 1.0), length_penalty=float(length_penalty),
                        early_stopping=int(bool(early_stopping)), exp_noise=noise.data_ptr() if noise is not None else None, seed=seed)
        n = ctypes.c_int(0)
        _lib.check(lib.idxtts_gpt_generate_beam(self._h, _lib.ptr(emb), pad_left.ctypes.data_as(c_void_p), B, P, max_new_tokens,
                                                float(repetition_penalty), ctypes.byref(bc), _lib.ptr(codes), ctypes.byref(n), _lib.ptr(ws),
                                                ws.numel(), int(use_graph), _lib.current_stream()))
        return torch.cat([input_ids.to(self.device), codes[:, : n.value]], dim=1)

    def inference_speech(self, speech_condition, text_inputs, emo_speech_condition=None, cond_lengths=None, emo_cond_lengths=None,
                         emo_vec=None, use_speed=False, input_tokens=None, num_return_sequences=1, max_generate_length=None,
                         typical_sampling=False, typical_mass=.9, return_logits=False, **hf_generate_kwargs):
        """`inference_speech` (model_v2.py:796-895): returns (codes [B, n], speech_conditioning_latent).
        speech_condition: the prompt features [B, T, 1024] as the reference takes them (conditioning encoders run here, model_v2.py:
        819-828) or -- hoisted out of the caller's segment loop -- the ready conditioning latent [B, 32, d] together with `emo_vec`.
        Decoding modes = HF generate's: greedy (do_sample=False, num_beams=1), multinomial sampling with the warpers
        (do_sample=True, num_beams=1; extra kwargs `exp_noise` / `generator` / `sampler`, see generate()), beam search and
        beam-sample (num_beams > 1: generate_beam())."""
        if input_tokens is not None or typical_sampling or use_speed:
            raise NotImplementedError("input_tokens / typical_sampling / use_speed are not used by IndexTTS2.infer and not implemented")
        if num_return_sequences != 1:
            raise NotImplementedError("num_return_sequences must be 1")
        sc = speech_condition if speech_condition.ndim == 3 else speech_condition.unsqueeze(0)
        if sc.shape[-1] == self.cfg.cond_module.input_size and sc.shape[-1] != self.cfg.model_dim:      # raw features
            if emo_speech_condition is None:
                emo_speech_condition = sc
            speech_conditioning_latent = self.get_conditioning(sc.transpose(1, 2), cond_lengths)
            if emo_vec is None:
                emo_vec = self.get_emovec(emo_speech_condition, emo_cond_lengths)                      # model_v2.py:823-826
        else:
            speech_conditioning_latent = sc
            if emo_vec is None:
                raise ValueError("pass emo_vec together with a ready conditioning latent")
        do_sample = bool(hf_generate_kwargs.pop("do_sample", False))
        num_beams = int(hf_generate_kwargs.pop("num_beams", 1))
        penalty = float(hf_generate_kwargs.pop("repetition_penalty", 1.0))
        samp = {"do_sample": do_sample, "top_p": hf_generate_kwargs.pop("top_p", 1.0), "top_k": hf_generate_kwargs.pop("top_k", 50),
                "temperature": hf_generate_kwargs.pop("temperature", 1.0), "exp_noise": hf_generate_kwargs.pop("exp_noise", None),
                "generator": hf_generate_kwargs.pop("generator", None)}
        forced_codes = hf_generate_kwargs.pop("forced_codes", None)
        sampler = hf_generate_kwargs.pop("sampler", "hf")
        length_penalty = float(hf_generate_kwargs.pop("length_penalty", 1.0))
        early_stopping = hf_generate_kwargs.pop("early_stopping", False)
        use_graph = bool(hf_generate_kwargs.pop("use_graph", True))
        if hf_generate_kwargs:
            raise TypeError(f"unsupported generate kwargs: {sorted(hf_generate_kwargs)}")
        B = text_inputs.shape[0]
        lat = speech_conditioning_latent.expand(B, -1, -1) if speech_conditioning_latent.shape[0] == 1 else speech_conditioning_latent
        ev = emo_vec.expand(B, -1) if emo_vec.shape[0] == 1 else emo_vec
        conds = self.conds_latent(lat, ev)
        input_ids, inputs_embeds, attention_mask = self.prepare_gpt_inputs(conds, text_inputs)
        trunc_index = input_ids.shape[1]
        max_new = (self.cfg.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
        if num_beams > 1:
            if return_logits:
                raise NotImplementedError("return_logits is a num_beams=1 feature")
            out = self.generate_beam(input_ids, max_new, attention_mask, inputs_embeds, num_beams=num_beams, repetition_penalty=penalty,
                                     length_penalty=length_penalty, early_stopping=early_stopping, use_graph=use_graph, **samp)
            return out[:, trunc_index:], speech_conditioning_latent
        out = self.generate(input_ids, max_new_tokens=max_new, stop_tokens=[self.cfg.stop_mel_token],
                            attention_mask=attention_mask, tts_embeddings=inputs_embeds, repetition_penalty=penalty,
                            return_logits=return_logits, sampler=sampler, use_graph=use_graph, forced_codes=forced_codes, **samp)
        if return_logits:
            return out[0][:, trunc_index:], speech_conditioning_latent, out[1]
        return out[:, trunc_index:], speech_conditioning_latent

    # ------------------------------------------------------------------------------------------
    def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, mel_codes_lengths,
                emo_speech_conditioning_latent=None, cond_mel_lengths=None, emo_cond_mel_lengths=None, emo_vec=None,
                use_speed=None, do_spk_cond=False) -> torch.Tensor:
        """Latent pass (model_v2.py:673-723) -> [B, M, d]."""
        if do_spk_cond or emo_vec is None:
            raise NotImplementedError("conditioning encoders are outside this hot path: pass the latent and emo_vec")
        cfg = self.cfg
        text = text_inputs.detach().cpu().numpy().astype(np.int64).copy()
        codes = mel_codes.detach().cpu().numpy().astype(np.int64).copy()
        B, L = text.shape
        M = codes.shape[1]
        tl = np.asarray(text_lengths.detach().cpu() if isinstance(text_lengths, torch.Tensor) else text_lengths).reshape(-1)
        ml = np.asarray(mel_codes_lengths.detach().cpu() if isinstance(mel_codes_lengths, torch.Tensor) else mel_codes_lengths).reshape(-1)
        tlv = [int(tl[min(b, len(tl) - 1)]) for b in range(B)]
        mlv = [int(ml[min(b, len(ml) - 1)]) for b in range(B)]
        for b in range(B):      # set_mel_padding (model_v2.py:569-581)
            codes[b, mlv[b]:] = cfg.stop_mel_token
        min_ = np.concatenate([np.full((B, 1), cfg.start_mel_token), codes, np.full((B, 1), cfg.stop_mel_token)], 1)
        conds = self.conds_latent(speech_conditioning_latent, emo_vec)
        nc = conds.shape[1]
        S = nc + L + 2 + M + 2
        tid = -np.ones((B, S), np.int32); tpos = -np.ones((B, S), np.int32)
        mid = -np.ones((B, S), np.int32); mpos = -np.ones((B, S), np.int32)
        eidx = -np.ones((B, S), np.int32)
        pad_left = np.zeros(B, np.int32)
        for b in range(B):
            # row layout [pad][conds][start, text_b, stop][start, codes, stop, stop]; a row whose text is shorter than L is
            # LEFT-padded (and the pad masked), which reproduces the reference's unpadded B=1 call for that row
            pad = L - tlv[b]
            pad_left[b] = pad
            trow = np.concatenate([[cfg.start_text_token], text[b, :tlv[b]], [cfg.stop_text_token]])
            eidx[b, pad:pad + nc] = np.arange(nc) + b * nc
            tid[b, pad + nc:nc + L + 2] = trow
            tpos[b, pad + nc:nc + L + 2] = np.arange(tlv[b] + 2)
            mid[b, nc + L + 2:] = min_[b]
            mpos[b, nc + L + 2:] = np.arange(M + 2)
        emb = self._embed(B * S, tid.reshape(-1), tpos.reshape(-1), mid.reshape(-1), mpos.reshape(-1),
                          conds.reshape(-1, cfg.model_dim), eidx.reshape(-1))
        latent = torch.empty(B, M, cfg.model_dim, device=self.device, dtype=torch.float32)
        ws = self._workspace(B, S, 0)
        pl = pad_left.ctypes.data_as(c_void_p) if pad_left.any() else c_void_p(0)
        _lib.check(_lib.load().idxtts_gpt_latent(self._h, _lib.ptr(emb), pl, B, S, nc + L + 2, M, _lib.ptr(latent), _lib.ptr(ws),
                                                 ws.numel(), _lib.current_stream()))
        return latent

    __call__ = forward

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().idxtts_ctx_destroy(self._h)
        except Exception:
            pass
