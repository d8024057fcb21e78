"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the IndexTTS-2 GPT stage.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

What it restates (reference file:line, relative to grantjr1842/index-tts):
  * GPT-2 block arithmetic      in-tree spec indextts/gpt/transformers_gpt2.py:196-234 (attention),
                                578-592 (MLP, gelu_new), 615-674 (block), 1171 (ln_f); the module
                                actually instantiated is third-party `transformers==4.52.1` GPT2Model
                                (model_v2.py:290-305; wpe nulled 300-302)
  * prepare_gpt_inputs          indextts/gpt/model_v2.py:725-794
  * GPT2InferenceModel.forward  model_v2.py:131-225 (prefill 162-172, decode 173-177: the mel position
                                index is attention_mask.shape[1] - mel_len, i.e. 0, 2, 3, 4, ...)
  * greedy loop                 indextts/gpt/transformers_generation_utils.py:3196-3269 with the
                                RepetitionPenalty processor (900-901; HF semantics: score<0 ? score*p : score/p
                                over every id in input_ids, including the fake all-ones prefix and 8192)
  * latent pass                 UnifiedVoice.forward model_v2.py:673-723, get_logits 597-625
PARITY PIN: the reference's own tests hold no numeric fixture for this stage and model_v2.py cannot be
imported here (transformers 5.x vs pinned 4.52.1), so the block arithmetic is pinned against the
container's `transformers.GPT2Model` + `RepetitionPenaltyLogitsProcessor` (tests/golden/make_golden.py
-> gpt.npz), the wrapper semantics against the text of model_v2.py cited above.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, key) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def gpt2_stack(w, cfg, emb: torch.Tensor, key_valid: Optional[torch.Tensor] = None,
               past: Optional[list] = None, kv_round: bool = False) -> Tuple[torch.Tensor, list]:
    """emb [B,S,d] (already includes the learned positions; HF wpe is nulled) -> ln_f(hidden) [B,S,d].

    key_valid: [B, past_len+S] bool/0-1 (the HF `attention_mask`), None = all valid.
    past: per-layer (K,V) each [B,H,past_len,hd].  Returns (hidden, present).
    kv_round: the bf16 KV-cache mode of the HIP path (include/idxtts.h::idxtts_gpt_set_kv_format; the reference's `use_fp16`
    keeps `past_key_values` in half precision, infer_v2.py:145-146): every key / value is rounded to bf16 (nearest even) when it
    is produced, prefill and decode alike, and used at that value from then on; all arithmetic stays fp32."""
    B, S, d = emb.shape
    H, hd = cfg.heads, cfg.head_dim
    past_len = 0 if past is None else past[0][0].shape[2]
    total = past_len + S
    qpos = torch.arange(past_len, total)[:, None]
    kpos = torch.arange(total)[None, :]
    allowed = (kpos <= qpos)[None, None]                                   # causal [1,1,S,total]
    if key_valid is not None:
        allowed = allowed & key_valid.bool()[:, None, None, :]
    x = emb
    present = []
    for i in range(cfg.layers):
        p = f"gpt.h.{i}"
        h = F.layer_norm(x, (d,), _t(w, f"{p}.ln_1.weight"), _t(w, f"{p}.ln_1.bias"), 1e-5)
        qkv = h @ _t(w, f"{p}.attn.c_attn.weight") + _t(w, f"{p}.attn.c_attn.bias")
        q, k, v = qkv.split(d, dim=2)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        if kv_round:
            k, v = k.to(torch.bfloat16).to(torch.float32), v.to(torch.bfloat16).to(torch.float32)
        if past is not None:
            k = torch.cat([past[i][0], k], dim=2)
            v = torch.cat([past[i][1], v], dim=2)
        present.append((k, v))
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        scores = torch.where(allowed, scores, torch.full_like(scores, torch.finfo(scores.dtype).min))
        att = torch.softmax(scores, dim=-1)
        a = (att @ v).transpose(1, 2).reshape(B, S, d)
        a = a @ _t(w, f"{p}.attn.c_proj.weight") + _t(w, f"{p}.attn.c_proj.bias")
        x = x + a
        h = F.layer_norm(x, (d,), _t(w, f"{p}.ln_2.weight"), _t(w, f"{p}.ln_2.bias"), 1e-5)
        m = gelu_new(h @ _t(w, f"{p}.mlp.c_fc.weight") + _t(w, f"{p}.mlp.c_fc.bias"))
        m = m @ _t(w, f"{p}.mlp.c_proj.weight") + _t(w, f"{p}.mlp.c_proj.bias")
        x = x + m
    x = F.layer_norm(x, (d,), _t(w, "gpt.ln_f.weight"), _t(w, "gpt.ln_f.bias"), 1e-5)
    return x, present


def lm_head(w, cfg, hidden: torch.Tensor) -> torch.Tensor:
    """lm_head = Sequential(final_norm, mel_head) applied AFTER ln_f (model_v2.py:208, 562)."""
    d = cfg.model_dim
    h = F.layer_norm(hidden, (d,), _t(w, "final_norm.weight"), _t(w, "final_norm.bias"), 1e-5)
    return h @ _t(w, "mel_head.weight").t() + _t(w, "mel_head.bias")


def conds_latent(w, cfg, speech_conditioning_latent: torch.Tensor, emo_vec: torch.Tensor) -> torch.Tensor:
    """model_v2.py:830-834: cat(latent + emo_vec, speed_emb(1), speed_emb(0)) -> [B, 34, d]."""
    B = speech_conditioning_latent.shape[0]
    se = _t(w, "speed_emb.weight")
    return torch.cat([speech_conditioning_latent + emo_vec[:, None, :],
                      se[1][None, None, :].expand(B, 1, -1), se[0][None, None, :].expand(B, 1, -1)], dim=1)


def prepare_gpt_inputs(w, cfg, conds: torch.Tensor, text_inputs: torch.Tensor):
    """model_v2.py:725-794 -> (fake_inputs [B,P+1], inputs_embeds [B,P,d], attention_mask [B,P+1])."""
    B, L = text_inputs.shape
    single = conds.shape[0] == 1
    target_len = conds.shape[1] + L + 2
    te, tp = _t(w, "text_embedding.weight"), _t(w, "text_pos_embedding.emb.weight")
    embs, masks = [], []
    for i in range(B):
        ti = text_inputs[i]
        ti = ti[(ti != cfg.stop_text_token) & (ti != cfg.start_text_token)]
        ti = F.pad(F.pad(ti, (1, 0), value=cfg.start_text_token), (0, 1), value=cfg.stop_text_token)
        temb = te[ti.long()] + tp[: ti.shape[0]]
        parts = [conds[0] if single else conds[i], temb]
        mask = torch.ones(target_len + 1, dtype=torch.long)
        padding = L + 2 - ti.shape[0]
        if padding > 0:
            parts.insert(0, torch.zeros(padding, conds.shape[-1]))
            mask[:padding] = 0
        embs.append(torch.cat(parts))
        masks.append(mask)
    inputs_embeds = torch.stack(embs)
    attention_mask = torch.stack(masks)
    fake = torch.ones(B, target_len + 1, dtype=torch.long)
    fake[:, -1] = cfg.start_mel_token
    return fake, inputs_embeds, attention_mask


def repetition_penalty(input_ids: torch.Tensor, scores: torch.Tensor, penalty: float) -> torch.Tensor:
    """HF RepetitionPenaltyLogitsProcessor: gather, rescale, scatter."""
    s = torch.gather(scores, 1, input_ids)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, input_ids, s)


def generate_greedy(w, cfg, conds: torch.Tensor, text_inputs: torch.Tensor, max_new_tokens: int,
                    repetition_penalty_value: float = 10.0, return_logits: bool = False, kv_round: bool = False):
    """inference_speech (model_v2.py:835-892) with do_sample=False, num_beams=1.
    Returns codes [B, n_steps] (eos and post-eos pad = stop_mel_token included, as HF returns them)."""
    fake, inputs_embeds, attention_mask = prepare_gpt_inputs(w, cfg, conds, text_inputs)
    B, P, d = inputs_embeds.shape
    me, mp = _t(w, "mel_embedding.weight"), _t(w, "mel_pos_embedding.emb.weight")
    input_ids = fake.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    past = None
    all_logits = []
    for step in range(max_new_tokens):
        if past is None:     # prefill: cached [pad|cond|text] embeddings + start_mel at mel position 0
            start = (me[cfg.start_mel_token] + mp[0])[None, None, :].expand(B, 1, d)
            emb = torch.cat([inputs_embeds, start], dim=1)
        else:                # decode: position = attention_mask.shape[1] - mel_len  (model_v2.py:175-177)
            pos = attention_mask.shape[1] - P
            emb = (me[input_ids[:, -1]] + mp[pos])[:, None, :]
        hidden, past = gpt2_stack(w, cfg, emb, attention_mask, past, kv_round)
        logits = lm_head(w, cfg, hidden[:, -1]).float()
        if return_logits:
            all_logits.append(logits.clone())
        scores = repetition_penalty(input_ids, logits, repetition_penalty_value) if repetition_penalty_value != 1.0 else logits
        nxt = torch.argmax(scores, dim=-1)
        nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=1)
        attention_mask = torch.cat([attention_mask, torch.ones(B, 1, dtype=torch.long)], dim=1)
        unfinished = unfinished & (nxt != cfg.stop_mel_token).long()
        if unfinished.max() == 0:
            break
    codes = input_ids[:, P + 1:]
    return (codes, torch.stack(all_logits, 1)) if return_logits else codes


def warp_scores(scores: torch.Tensor, temperature: float, top_k: int, top_p: float, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """The sampling warpers HF generate() applies after the repetition penalty, in its order (reference:
    transformers_generation_utils.py:1036-1044 builds TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper from
    the pinned third-party `transformers` 4.52.1 logits_process.py; num_beams=1 gives min_tokens_to_keep=1, line 1031):
      temperature: scores / T
      top-k: remove everything strictly below the k-th largest value (ties at the threshold stay)
      top-p: sort ascending, softmax, cumulative sum; remove where cum <= 1 - top_p, never the last min_tokens_to_keep."""
    if temperature != 1.0:
        scores = scores / temperature
    if top_k and top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), scores.shape[-1])
        kth = torch.topk(scores, k)[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(scores, descending=False)
        cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = False
        scores = scores.masked_fill(remove.scatter(1, sorted_indices, remove), float("-inf"))
    return scores


def generate_sample(w, cfg, conds: torch.Tensor, text_inputs: torch.Tensor, max_new_tokens: int, exp_noise: torch.Tensor,
                    repetition_penalty_value: float = 10.0, temperature: float = 0.8, top_k: int = 30, top_p: float = 0.8,
                    accel_sampler: bool = False, kv_round: bool = False):
    """inference_speech (model_v2.py:835-892) with do_sample=True, num_beams=1: HF _sample
    (transformers_generation_utils.py:3196-3262): repetition penalty -> warpers -> softmax -> torch.multinomial(probs, 1).
    torch.multinomial with one sample per row IS argmax(probs / q), q ~ Exp(1) drawn by ONE exponential_() call on a
    [B, V] tensor (ATen native multinomial, fast path), so the draw is an explicit input here: exp_noise [steps, B, V].
    accel_sampler=True restates the accel engine's Sampler instead (accel_engine.py:648-659): softmax(logits / T),
    divided by clamp_min(q, 1e-10), argmax -- no repetition penalty, no top-k / top-p."""
    fake, inputs_embeds, attention_mask = prepare_gpt_inputs(w, cfg, conds, text_inputs)
    B, P, d = inputs_embeds.shape
    me, mp = _t(w, "mel_embedding.weight"), _t(w, "mel_pos_embedding.emb.weight")
    input_ids = fake.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    past = None
    for step in range(max_new_tokens):
        if past is None:
            start = (me[cfg.start_mel_token] + mp[0])[None, None, :].expand(B, 1, d)
            emb = torch.cat([inputs_embeds, start], dim=1)
        else:
            pos = attention_mask.shape[1] - P
            emb = (me[input_ids[:, -1]] + mp[pos])[:, None, :]
        hidden, past = gpt2_stack(w, cfg, emb, attention_mask, past, kv_round)
        logits = lm_head(w, cfg, hidden[:, -1]).float()
        q = exp_noise[step].float()
        if accel_sampler:
            probs = torch.softmax(logits / temperature, dim=-1)
            nxt = (probs / q.clamp_min(1e-10)).argmax(dim=-1)
        else:
            scores = repetition_penalty(input_ids, logits, repetition_penalty_value) if repetition_penalty_value != 1.0 else logits
            scores = warp_scores(scores, temperature, top_k, top_p)
            probs = torch.softmax(scores, dim=-1)
            nxt = (probs / q).argmax(dim=-1)
        nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=1)
        attention_mask = torch.cat([attention_mask, torch.ones(B, 1, dtype=torch.long)], dim=1)
        unfinished = unfinished & (nxt != cfg.stop_mel_token).long()
        if unfinished.max() == 0:
            break
    return input_ids[:, P + 1:]


class _BeamHyps:
    """BeamHypotheses (indextts/gpt/transformers_beam_search.py:930-1013): n-best list of finished hypotheses."""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping: bool):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams = []
        self.worst_score = 1e9

    def add(self, hyp: torch.Tensor, sum_logprobs: float, generated_len: int):
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                ranked = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int, prompt_len: int) -> bool:
        if len(self.beams) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        return self.worst_score >= best_sum_logprobs / (cur_len - prompt_len) ** self.length_penalty


def generate_beam(w, cfg, conds: torch.Tensor, text_inputs: torch.Tensor, max_new_tokens: int, exp_noise: Optional[torch.Tensor],
                  num_beams: int = 3, repetition_penalty_value: float = 10.0, temperature: float = 0.8, top_k: int = 30,
                  top_p: float = 0.8, length_penalty: float = 0.0, do_sample: bool = True, early_stopping: bool = False,
                  return_trace: bool = False, kv_round: bool = False):
    """inference_speech (model_v2.py:835-892) with num_beams > 1 -- the mode `IndexTTS2.infer` really runs by default
    (infer_v2.py:714-722, 767: do_sample=True, num_beams=3, top_p=.8, top_k=30, temperature=.8, repetition_penalty=10,
    length_penalty=0).  Restates the vendored `GenerationMixin._beam_search` (transformers_generation_utils.py:3325-3516):
      log_softmax(fp32 logits) BEFORE the processors (3473-3477); processors = repetition penalty then, when sampling, the
      warpers with min_tokens_to_keep = 2 (1022-1029); + running beam scores (first beam 0, the others -1e9: 3420-3422);
      2 * num_beams candidates over the num_beams * V flattened scores -- `torch.multinomial` without replacement
      (3509-3510: ATen draws q ~ Exp(1) ONCE for the [B, num_beams * V] tensor and takes topk(probs / q), so the draw is the
      explicit input exp_noise[step]) re-sorted by score (3511-3513), or plain top-k when do_sample=False (3517-3519);
      BeamSearchScorer.process / finalize (transformers_beam_search.py:215-318, 320-414) with BeamHypotheses above; the KV
      cache re-indexed by beam_idx every step (`_reorder_cache`, model_v2.py:227-240).
    Returns codes [B, n] (best hypothesis per utterance, eos appended when it fits, stop-token padded)."""
    fake, inputs_embeds, attention_mask = prepare_gpt_inputs(w, cfg, conds, text_inputs)
    B, P, d = inputs_embeds.shape
    nb, V = num_beams, cfg.number_mel_codes
    eos = pad = cfg.stop_mel_token
    me, mp = _t(w, "mel_embedding.weight"), _t(w, "mel_pos_embedding.emb.weight")
    input_ids = fake.repeat_interleave(nb, 0)                       # _expand_inputs_for_generation
    attention_mask = attention_mask.repeat_interleave(nb, 0)
    embeds = inputs_embeds.repeat_interleave(nb, 0)                 # model_v2.py:166-169
    prompt_len = input_ids.shape[1]
    max_length = prompt_len + max_new_tokens
    beam_scores = torch.zeros(B, nb)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(nb, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    past = None
    step = 0
    trace = []
    while True:
        if past is None:
            start = (me[cfg.start_mel_token] + mp[0])[None, None, :].expand(B * nb, 1, d)
            emb = torch.cat([embeds, start], dim=1)
        else:
            emb = (me[input_ids[:, -1]] + mp[attention_mask.shape[1] - P])[:, None, :]
        hidden, past = gpt2_stack(w, cfg, emb, attention_mask, past, kv_round)
        logits = lm_head(w, cfg, hidden[:, -1]).float()
        scores = F.log_softmax(logits, dim=-1)
        proc = repetition_penalty(input_ids, scores, repetition_penalty_value) if repetition_penalty_value != 1.0 else scores
        if do_sample:
            proc = warp_scores(proc, temperature, top_k, top_p, min_tokens_to_keep=2)
        scores = (proc + beam_scores[:, None]).view(B, nb * V)
        n_keep = 2 * nb
        if do_sample:
            probs = F.softmax(scores, dim=-1)
            key = probs / exp_noise[step].float()
            cand = torch.topk(key, n_keep, dim=-1)[1]       # == torch.multinomial(probs, n_keep)
            cand_scores = torch.gather(scores, -1, cand)
            cand_scores, order = torch.sort(cand_scores, descending=True, dim=1)
            cand = torch.gather(cand, -1, order)
        else:
            key = scores
            cand_scores, cand = torch.topk(scores, n_keep, dim=1, largest=True, sorted=True)
        if return_trace:
            # how close this step's decisions are: the smallest gap between neighbours among the best n_keep + 1 selection keys (who is
            # drawn), relative for the sampling key probs / q, and among the drawn candidates' scores (their order = the beams' order)
            kt = torch.topk(key, n_keep + 1, dim=-1)[0]
            gsel = (kt[:, :-1] - kt[:, 1:]) / (kt[:, :-1].abs().clamp_min(1e-30) if do_sample else 1.0)
            gord = cand_scores[:, :-1] - cand_scores[:, 1:]
            step_gap = torch.minimum(gsel.min(dim=1)[0], gord.min(dim=1)[0])
        cand_beam = torch.div(cand, V, rounding_mode="floor")
        cand_tok = cand % V
        # ---- BeamSearchScorer.process ----
        cur_len = input_ids.shape[-1] + 1
        nxt_scores = torch.zeros(B, nb)
        nxt_tok = torch.zeros(B, nb, dtype=torch.long)
        nxt_idx = torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nxt_tok[b, :] = pad
                continue
            slot = 0
            for rank in range(n_keep):
                tok, sc, src = int(cand_tok[b, rank]), float(cand_scores[b, rank]), b * nb + int(cand_beam[b, rank])
                if tok == eos:
                    if rank >= nb:
                        continue
                    hyps[b].add(input_ids[src].clone(), sc, cur_len - prompt_len)
                else:
                    nxt_scores[b, slot], nxt_tok[b, slot], nxt_idx[b, slot] = sc, tok, src
                    slot += 1
                if slot == nb:
                    break
            assert slot == nb, "fewer than num_beams non-eos candidates"
            done[b] = done[b] or hyps[b].is_done(float(cand_scores[b].max()), cur_len, prompt_len)
        beam_scores = nxt_scores.view(-1)
        beam_idx = nxt_idx.view(-1)
        if return_trace:
            trace.append((beam_idx.clone(), nxt_tok.view(-1).clone(), beam_scores.clone(), step_gap.clone()))
        input_ids = torch.cat([input_ids[beam_idx], nxt_tok.view(-1, 1)], dim=-1)
        past = [(k.index_select(0, beam_idx), v.index_select(0, beam_idx)) for k, v in past]
        attention_mask = torch.cat([attention_mask, torch.ones(B * nb, 1, dtype=torch.long)], dim=1)
        step += 1
        if all(done) or input_ids.shape[-1] >= max_length:
            break
    # ---- BeamSearchScorer.finalize ----
    for b in range(B):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(input_ids[b * nb + j], float(beam_scores[b * nb + j]), input_ids.shape[-1] - prompt_len)
    best = [sorted(h.beams, key=lambda x: x[0]).pop()[1] for h in hyps]
    lens = [len(x) for x in best]
    sent_max = min(max(lens) + 1, max_length)
    decoded = torch.full((B, sent_max), pad, dtype=torch.long)
    for b, hyp in enumerate(best):
        decoded[b, :lens[b]] = hyp
        if lens[b] < sent_max:
            decoded[b, lens[b]] = eos
    codes = decoded[:, prompt_len:]
    return (codes, trace) if return_trace else codes


def latent_forward(w, cfg, speech_conditioning_latent: torch.Tensor, text_inputs: torch.Tensor,
                   mel_codes: torch.Tensor, emo_vec: torch.Tensor,
                   text_lengths: Optional[torch.Tensor] = None,
                   mel_lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UnifiedVoice.forward (model_v2.py:673-723) with do_spk_cond=False, emo_vec given -> latent [B,M,d]."""
    B, L = text_inputs.shape
    M = mel_codes.shape[1]
    text_inputs = text_inputs.clone().long()
    mel_codes = mel_codes.clone().long()
    if text_lengths is not None:       # set_text_padding (model_v2.py:583-595)
        for b in range(B):
            text_inputs[b, int(text_lengths[b]):] = cfg.stop_text_token
    if mel_lengths is not None:        # set_mel_padding (569-581)
        for b in range(B):
            mel_codes[b, int(mel_lengths[b]):] = cfg.stop_mel_token
    text_inputs = F.pad(text_inputs, (0, 1), value=cfg.stop_text_token)
    mel_codes = F.pad(mel_codes, (0, 1), value=cfg.stop_mel_token)
    conds = conds_latent(w, cfg, speech_conditioning_latent, emo_vec)
    text_in = F.pad(text_inputs, (1, 0), value=cfg.start_text_token)      # build_aligned_inputs_and_targets
    mel_in = F.pad(mel_codes, (1, 0), value=cfg.start_mel_token)
    text_emb = _t(w, "text_embedding.weight")[text_in] + _t(w, "text_pos_embedding.emb.weight")[: L + 2]
    mel_emb = _t(w, "mel_embedding.weight")[mel_in] + _t(w, "mel_pos_embedding.emb.weight")[: M + 2]
    emb = torch.cat([conds, text_emb, mel_emb], dim=1)
    hidden, _ = gpt2_stack(w, cfg, emb)
    enc = F.layer_norm(hidden[:, conds.shape[1]:], (cfg.model_dim,), _t(w, "final_norm.weight"),
                       _t(w, "final_norm.bias"), 1e-5)
    mel_part = enc[:, -(M + 2):]
    return mel_part[:, :-2]
