"""CPU: the GPT oracle (oracle/gpt.py) against fixtures produced with the container's HF GPT2Model
(tests/golden/make_golden.py::make_gpt)."""
import os

import numpy as np
import torch

from indextts_amd import synth, weights
from indextts_amd.config import GPTConfig
from oracle import gpt as og


def _setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt.npz"))
    cfg = GPTConfig.tiny()
    w = {k: torch.from_numpy(v) for k, v in weights.synth_gpt_weights(cfg, tag="golden/gpt").items()}
    return g, cfg, w


def test_greedy_codes_and_logits_match_hf(golden_dir):
    g, cfg, w = _setup(golden_dir)
    B, L = g["greedy_text"].shape
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5))
    text = torch.from_numpy(g["greedy_text"])
    codes, logits = og.generate_greedy(w, cfg, conds, text, max_new_tokens=g["greedy_codes"].shape[1], return_logits=True)
    assert np.array_equal(codes.numpy(), g["greedy_codes"])          # token indices: bit-exact
    np.testing.assert_allclose(logits.numpy(), g["greedy_logits"], rtol=0, atol=2e-4)
    assert np.abs(g["greedy_logits"]).max() > 1.0


def test_prepare_inputs_left_pads_ragged_rows(golden_dir):
    g, cfg, w = _setup(golden_dir)
    B, L = g["greedy_text"].shape
    conds = torch.zeros(1, cfg.cond_latents + 2, cfg.model_dim)
    fake, emb, mask = og.prepare_gpt_inputs(w, cfg, conds, torch.from_numpy(g["greedy_text"]))
    P = cfg.cond_latents + 2 + L + 2
    assert fake.shape == (B, P + 1) and emb.shape == (B, P, cfg.model_dim) and mask.shape == (B, P + 1)
    assert (fake[:, :-1] == 1).all() and (fake[:, -1] == cfg.start_mel_token).all()
    assert mask[0].sum() == P + 1 and mask[1].sum() == P + 1 - 3 and mask[2].sum() == P + 1 - 7
    assert (emb[2, :7] == 0).all() and (mask[2, :7] == 0).all()


def test_padding_invariance_of_greedy_codes(golden_dir):
    """The reference's own property test (tests/padding_test.py:35-89): a row decoded alone gives the
    same greedy codes as the same row left-padded inside a batch."""
    g, cfg, w = _setup(golden_dir)
    B, L = g["greedy_text"].shape
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5))
    text = torch.from_numpy(g["greedy_text"])
    solo = og.generate_greedy(w, cfg, conds[2:3], text[2:3, :5], max_new_tokens=8)
    assert np.array_equal(solo.numpy()[0], g["greedy_codes"][2, :8])


def test_latent_pass_matches_hf(golden_dir):
    g, cfg, w = _setup(golden_dir)
    B, M, d = g["latent"].shape
    lat = torch.from_numpy(synth.uniform("golden/gpt/lat", (B, cfg.cond_latents, d), 0.5))
    emo = torch.from_numpy(synth.uniform("golden/gpt/emo", (B, d), 0.3))
    text = torch.from_numpy(synth.integers("golden/gpt/text2", (B, 7), 2, cfg.number_text_tokens))
    codes = torch.from_numpy(synth.integers("golden/gpt/codes2", (B, M), 0, cfg.start_mel_token))
    out = og.latent_forward(w, cfg, lat, text, codes, emo)
    np.testing.assert_allclose(out.numpy(), g["latent"], rtol=0, atol=2e-5)


def test_eos_bookkeeping():
    """Rows that emit the stop token keep emitting it (pad = eos = stop_mel_token) until all rows finish
    (transformers_generation_utils.py:3255-3264)."""
    cfg = GPTConfig.tiny()
    w = {k: torch.from_numpy(v) for k, v in weights.synth_gpt_weights(cfg, tag="t/gpt/eos").items()}
    w["mel_head.bias"] = w["mel_head.bias"].clone()
    w["mel_head.bias"][cfg.stop_mel_token] = 3.5   # make EOS likely early for some rows
    conds = torch.from_numpy(synth.uniform("t/gpt/eos/conds", (4, cfg.cond_latents + 2, cfg.model_dim), 0.5))
    text = torch.from_numpy(synth.integers("t/gpt/eos/text", (4, 6), 2, cfg.number_text_tokens))
    codes = og.generate_greedy(w, cfg, conds, text, max_new_tokens=40).numpy()
    for row in codes:
        hits = np.nonzero(row == cfg.stop_mel_token)[0]
        if len(hits):
            assert (row[hits[0]:] == cfg.stop_mel_token).all()
    assert codes.shape[1] <= 40


def test_multinomial_sampling_matches_hf_golden(golden_dir):
    """do_sample=True, num_beams=1: the golden codes come from HF's own warpers + torch.multinomial under a fixed seed; the
    fixture also holds the Exp(1) draws multinomial consumed, so the oracle's argmax(probs / q) restatement must reproduce
    the tokens exactly."""
    g = np.load(os.path.join(golden_dir, "gpt.npz"))
    cfg = GPTConfig.tiny()
    tw = {k: torch.from_numpy(v) for k, v in weights.synth_gpt_weights(cfg, tag="golden/gpt").items()}
    B = g["greedy_text"].shape[0]
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5))
    text = torch.from_numpy(g["greedy_text"])
    temp, top_k, top_p = g["sample_params"]
    noise = torch.from_numpy(g["sample_noise"])
    codes = og.generate_sample(tw, cfg, conds, text, g["sample_codes"].shape[1], noise, 10.0, float(temp), int(top_k), float(top_p))
    assert np.array_equal(codes.numpy(), g["sample_codes"][:, :codes.shape[1]])
    assert not np.array_equal(g["sample_codes"], g["greedy_codes"])        # the fixture actually samples


def test_warp_scores_edge_cases():
    """top-k keeps ties at the threshold; top-p never removes the largest; disabled filters are identities."""
    s = torch.tensor([[1.0, 3.0, 3.0, 2.0, -1.0, 0.5]])
    w = og.warp_scores(s.clone(), 1.0, 2, 1.0)
    assert torch.isinf(w[0, [0, 3, 4, 5]]).all() and (w[0, [1, 2]] == 3.0).all()
    w = og.warp_scores(s.clone(), 1.0, 0, 1e-6)
    assert torch.isfinite(w).sum() >= 1 and torch.isfinite(w[0, 1:3]).any()
    assert torch.equal(og.warp_scores(s.clone(), 1.0, 0, 1.0), s)
    assert torch.allclose(og.warp_scores(s.clone(), 0.5, 0, 1.0), s / 0.5)


def test_kv_round_mode_rounds_keys_and_values_once_and_only_them(golden_dir):
    """The bf16 KV-cache mode of the oracle (gpt2_stack(kv_round=True), the checker of idxtts_gpt_set_kv_format): every cached key /
    value is a bf16 number (16 low mantissa bits zero), a cached step reads exactly what the prefill stored, the logits move by no
    more than bf16 rounding of the keys and values can explain, and the default mode is untouched (HF fixture above)."""
    g, cfg, w = _setup(golden_dir)
    B, S = 2, 9
    emb = torch.from_numpy(synth.uniform("t/oracle/kvround/emb", (B, S, cfg.model_dim), 1.0))
    with torch.no_grad():
        h32, past32 = og.gpt2_stack(w, cfg, emb[:, :-1])
        h16, past16 = og.gpt2_stack(w, cfg, emb[:, :-1], kv_round=True)
        for (k, v), (k0, v0) in zip(past16, past32):
            for t, t0 in ((k, k0), (v, v0)):
                bits = t.contiguous().view(torch.int32)
                assert int((bits & 0xFFFF).abs().max()) == 0
                assert (t - t0).abs().max().item() <= 2.0 ** -8 * t0.abs().max().item()
        step16, past16b = og.gpt2_stack(w, cfg, emb[:, -1:], past=past16, kv_round=True)
        full16, _ = og.gpt2_stack(w, cfg, emb, kv_round=True)
    for (k, v), (kp, vp) in zip(past16b, past16):
        assert torch.equal(k[:, :, :S - 1], kp) and torch.equal(v[:, :, :S - 1], vp)
    np.testing.assert_allclose(step16[:, 0].numpy(), full16[:, -1].numpy(), rtol=0, atol=2e-5)       # cached step == uncached row
    d = (h16 - h32).abs().max().item()
    assert 0 < d <= 0.05 * h32.abs().max().item()
