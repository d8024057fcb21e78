"""ORACLE-SIDE CHECKER (test infrastructure, not product): what "greedy codes bit-exact" means between two correct
implementations, measured instead of assumed.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Two fp32 implementations of the reference's greedy loop (transformers_generation_utils.py:3196-3269: logits -> repetition penalty
-> argmax) agree on a token exactly when their logits differ by less than the margin between the two best penalised scores.  With
reduced-precision STORAGE in the decode (the bf16 KV cache, the reference's `use_fp16`) the implementation noise in a logit grows,
and on random-init weights (nearly flat logits) a near-tie is met within a few dozen steps.  From the first flip on, a
free-running comparison compares two different utterances and says nothing.  So the parity statement is made TEACHER-FORCED:

  1. the oracle decodes freely:             codes_ref [B, n], logits_ref [B, n, V]
  2. the HIP path decodes the SAME tokens:  idxtts_gpt_generate_forced on codes_ref -> its logits and its own argmax per step
  3. report  max |logit_gpu - logit_ref|  (the implementation noise), the share of steps whose argmax agrees (match rate), and for
     EVERY disagreeing step the oracle's margin between its token and the HIP path's token -- which must be below twice the noise
     bound, or the disagreement is a bug and not a tie
  4. the free-running HIP decode must leave the oracle's sequence exactly at the first such step of each utterance (or never).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import gpt as og


def decode_parity(uv, tw: Dict[str, torch.Tensor], gcfg, spk_latent: torch.Tensor, emo_vec: torch.Tensor, text: torch.Tensor, n_codes: int,
                  penalty: float = 10.0) -> dict:
    """uv: indextts_amd.gpt.UnifiedVoice (the HIP path); tw: the state dict the oracle runs (for compact formats: the read-back rounded
    model, `uv.effective_state_dict`); spk_latent [B, 32, d], emo_vec [B, d], text [B, L].  The oracle rounds keys / values like the HIP
    path's cache (`kv_round`) when uv.kv_format == "bf16"."""
    kv_round = uv.kv_format == "bf16"
    B = text.shape[0]
    import time
    with torch.no_grad():
        conds = og.conds_latent(tw, gcfg, spk_latent, emo_vec)
        t0 = time.perf_counter()
        codes_ref, logits_ref = og.generate_greedy(tw, gcfg, conds, text, n_codes, penalty, return_logits=True, kv_round=kv_round)
        oracle_s = time.perf_counter() - t0
        fake = og.prepare_gpt_inputs(tw, gcfg, conds, text)[0]
    n = codes_ref.shape[1]
    forced = torch.full((B, n_codes), gcfg.stop_mel_token, dtype=torch.long)
    forced[:, :n] = codes_ref
    tf_codes, _, logits_gpu = uv.inference_speech(spk_latent, text, emo_vec=emo_vec, max_generate_length=n_codes, do_sample=False, num_beams=1,
                                                  repetition_penalty=penalty, return_logits=True, forced_codes=forced)
    tf_codes, logits_gpu = tf_codes.cpu()[:, :n], logits_gpu.cpu()[:, :n]
    free_codes, _ = uv.inference_speech(spk_latent, text, emo_vec=emo_vec, max_generate_length=n_codes, do_sample=False, num_beams=1,
                                        repetition_penalty=penalty)
    free_codes = free_codes.cpu()
    # Vocabulary entries the comparison is made on: the bench's fixed-length utterances carry a stop-token bias of -1e4, where the fp32
    # spacing alone is 1e-3 -- such pinned entries (|logit| >= 1e3) can never win the argmax and are left out of the noise figures.
    live = logits_ref.abs() < 1e3
    dl = (logits_gpu - logits_ref).abs() * live
    mism = tf_codes != codes_ref
    ties = []
    for b, s in zip(*np.nonzero(mism.numpy())):
        ids = torch.cat([fake[b], codes_ref[b, :s]])[None]
        sc = og.repetition_penalty(ids, logits_ref[b, s][None].float(), penalty)[0]
        top2 = torch.topk(sc, 2).values
        ties.append({"utterance": int(b), "step": int(s), "oracle_token": int(codes_ref[b, s]), "hip_token": int(tf_codes[b, s]),
                     "oracle_margin_to_hip_token": float(sc[int(codes_ref[b, s])] - sc[int(tf_codes[b, s])]),
                     "oracle_top2_margin": float(top2[0] - top2[1]), "abs_logit_diff_at_step": float(dl[b, s].max())})
    first_tf = [int(np.nonzero(mism[b].numpy())[0][0]) if mism[b].any() else None for b in range(B)]
    m = min(free_codes.shape[1], n)
    fdiff = free_codes[:, :m] != codes_ref[:, :m]
    first_free = [int(np.nonzero(fdiff[b].numpy())[0][0]) if fdiff[b].any() else None for b in range(B)]
    top2_all = []
    with torch.no_grad():
        for b in range(B):      # how flat the oracle's own decisions are: its margin between the two best penalised scores, every step
            for s in range(n):
                ids = torch.cat([fake[b], codes_ref[b, :s]])[None]
                t2 = torch.topk(og.repetition_penalty(ids, logits_ref[b, s][None].float(), penalty)[0], 2).values
                top2_all.append(float(t2[0] - t2[1]))
    return {
        "utterances": B, "steps": int(n), "kv_cache": uv.kv_format, "gpt_weights": uv.weight_format,
        "oracle_decode_seconds": round(oracle_s, 2), "oracle_decode_tokens_per_s": round(B * int(n) / oracle_s, 1),
        "max_abs_logit_diff": float(dl.max()), "mean_abs_logit_diff": float(dl.mean()), "logit_std": float(logits_ref[live].std()),
        "logit_entries_compared": "all with |oracle logit| < 1e3 (the -1e4 stop-token bias of fixed-length synthetic utterances is excluded)",
        "codes_match_rate_teacher_forced": float(1.0 - mism.float().mean()),
        "mismatching_steps": ties,
        "worst_oracle_margin_at_a_mismatch": max([t["oracle_margin_to_hip_token"] for t in ties], default=0.0),
        "oracle_top2_margin_median": float(np.median(top2_all)), "oracle_top2_margin_min": float(np.min(top2_all)),
        "first_mismatch_step_teacher_forced": first_tf, "first_difference_step_free_running": first_free,
        "free_running_codes_equal": bool(not fdiff.any()),
        "free_running_prefix_match_rate": float(np.mean([(first_free[b] if first_free[b] is not None else m) / m for b in range(B)])),
    }
