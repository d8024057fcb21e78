"""ORACLE (test infrastructure, not product): the reference's per-utterance flow GPT -> s2mel -> BigVGAN
(infer_v2.py:732-881, one segment, B = 1 as the reference runs it), assembled from the stage oracles.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch

from . import gpt as og
from . import s2mel as osm
from . import vocoder as ov


def synthesize_one(w_gpt, w_s2mel, w_voc, cfg, text_tokens: torch.Tensor, cond, noise: torch.Tensor, max_mel_tokens: int,
                   repetition_penalty: float = 10.0, diffusion_steps: int = None, cfg_rate: float = None, kv_round: bool = False,
                   timers: dict = None):
    """text_tokens [1, L] (no padding), cond = object with the PromptConditioning fields (CPU tensors),
    noise [1, 80, >= Tp + Tg].  kv_round: the bf16 KV-cache mode (oracle/gpt.py::gpt2_stack).  Returns dict(codes, latent, cond, mel, wav).
    timers (optional dict): receives the seconds of the three stages (gpt = greedy decode + latent pass, s2mel, bigvgan)."""
    import time
    t0 = time.perf_counter()
    g = cfg.gpt
    steps = cfg.diffusion_steps if diffusion_steps is None else diffusion_steps
    rate = cfg.cfg_rate if cfg_rate is None else cfg_rate
    conds = og.conds_latent(w_gpt, g, cond.spk_cond_latent, cond.emo_vec)
    codes = og.generate_greedy(w_gpt, g, conds, text_tokens, max_mel_tokens, repetition_penalty, kv_round=kv_round)   # infer_v2.py:760-777
    row = codes[0].numpy()
    hits = np.nonzero(row == g.stop_mel_token)[0]
    code_len = int(hits[0]) if len(hits) else len(row)                                                      # 795-807
    codes = codes[:, :code_len]
    latent = og.latent_forward(w_gpt, g, cond.spk_cond_latent, text_tokens, codes, cond.emo_vec)            # 816-828
    t1 = time.perf_counter()
    lat2 = osm.gpt_layer(w_s2mel, latent)                                                                   # 835
    S = osm.vq2emb(w_s2mel, codes) + lat2                                                                   # 841-843
    target = (torch.LongTensor([code_len]) * cfg.code_to_frame).long()                                      # 844
    c = osm.length_regulator(w_s2mel, cfg.s2mel, S, target)                                                 # 846
    cat = torch.cat([cond.prompt_condition, c], dim=1)                                                      # 850
    Tp, T = cond.ref_mel.shape[-1], cat.shape[1]
    mel = osm.cfm_inference(w_s2mel, cfg.s2mel, cat, torch.LongTensor([T]), cond.ref_mel, cond.style, noise[:, :, :T], steps, rate)
    vc = mel[:, :, Tp:]                                                                                     # 856
    t2 = time.perf_counter()
    wav = ov.bigvgan_forward(w_voc, cfg.bigvgan, vc.float())                                                # 860
    wav = torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0)                                            # 866
    if timers is not None:
        timers.update({"gpt": t1 - t0, "s2mel": t2 - t1, "bigvgan": time.perf_counter() - t2})
    return {"codes": codes, "code_len": code_len, "latent": latent, "cond": c, "mel": vc, "wav": wav}
