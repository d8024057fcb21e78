"""GPU, FULL-SIZE weights, at BASELINE.json's own sizes:
  configs[4]  long-form: one utterance, 1500 codes (30 s), emotion vector mixed in, fp8 GPT weight streams, graph-replayed decode
  configs[2]  batch 16 x 512 codes, greedy
The CPU oracle cannot run these end to end in test time, so: graph replay == eager launch for all 1500 steps (bit for bit),
the first codes and one estimator evaluation at the full T = 689 + 2580 against the oracle, the vocoder at 2580 frames
against the oracle on an interior window (locality), and batch rows == their own B = 1 runs."""
import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import PipelineConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(device):
    cfg = PipelineConfig()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4         # fixed-length synthetic utterances (as bench.py)
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
    return cfg, wg, ws, wv


def _cond(cfg, emo_scale=0.3):
    from indextts_amd.infer_v2 import PromptConditioning
    c = PromptConditioning.synthetic(cfg, prompt_frames=689, tag="bench/prompt")
    return c


def test_config4_longform_decode_graph_equals_eager_and_oracle(device, full):
    """1500 steps at B = 1 on fp8 weight streams: the hipGraph replay and the eager launches produce the same 1500 codes, and the
    first 24 equal the CPU oracle running the SAME rounded model (idxtts_ctx_get_tensor hands it back)."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg, wg, ws, wv = full
    uv = UnifiedVoice(wg, cfg.gpt, device=device, weight_format="fp8", keep_effective=True)
    c = _cond(cfg)
    emo2 = torch.from_numpy(synth.uniform("t/full/emo2", (1, cfg.gpt.model_dim), 0.3))
    emo = c.emo_vec + 1.0 * (emo2 - c.emo_vec)                  # merge_emovec's mix, alpha = 1.0 (configs[4]: emotion-prompted)
    text = torch.from_numpy(synth.integers("t/full/text4", (1, 128), 2, cfg.gpt.number_text_tokens))
    M = 1500
    runs = {}
    for graph in (True, False):
        codes, _ = uv.inference_speech(c.spk_cond_latent, text, emo_vec=emo, max_generate_length=M, do_sample=False, num_beams=1,
                                       repetition_penalty=10.0, use_graph=graph)
        runs[graph] = codes.cpu().numpy()
        assert runs[graph].shape == (1, M)
    assert np.array_equal(runs[True], runs[False])
    assert len(np.unique(runs[True])) > 50                      # not a degenerate loop
    tw = {k: torch.from_numpy(v) for k, v in uv.effective_state_dict.items()}
    ref = og.generate_greedy(tw, cfg.gpt, og.conds_latent(tw, cfg.gpt, c.spk_cond_latent, emo), text, 24, 10.0, kv_round=uv.kv_format == "bf16")
    assert uv.kv_format == "bf16" and np.array_equal(runs[True][:, :24], ref.numpy())


def test_config4_estimator_at_full_length_vs_oracle(device, full):
    """One DiT + WaveNet evaluation at T = 689 + 2580 frames (30 s of audio behind an 8 s prompt), full width, against the CPU oracle."""
    from indextts_amd import _lib
    from indextts_amd.s2mel import S2Mel
    from oracle import s2mel as osm
    cfg, wg, ws, wv = full
    sm = S2Mel(ws, cfg.s2mel, device=device)
    tws = {k: torch.from_numpy(v) for k, v in ws.items()}
    Tp, T = 689, 689 + 2580
    C = cfg.s2mel.in_channels
    x = torch.from_numpy(synth.uniform("t/full/dit/x", (1, C, T), 1.0))
    x[..., :Tp] = 0
    px = torch.zeros(1, C, T)
    px[..., :Tp] = torch.from_numpy(synth.uniform("t/full/dit/p", (1, C, Tp), 2.6, -4.0))
    st = torch.from_numpy(synth.uniform("t/full/dit/style", (1, cfg.s2mel.style_dim), 1.0))
    mu = torch.from_numpy(synth.uniform("t/full/dit/mu", (1, T, cfg.s2mel.content_dim), 1.0))
    t = torch.tensor([0.45])
    torch.set_num_threads(16)
    with torch.no_grad():
        want = osm.dit_forward(tws, cfg.s2mel, x, px, torch.LongTensor([T]), t, st, mu)
    scale = max(1.0, want.abs().max().item())
    try:
        for mode, tol in ((_lib.GEMM_F32, 2e-4), (_lib.GEMM_BF16X3, 1e-3)):
            _lib.set_gemm_mode(mode)
            got = sm.estimator(x, px, torch.LongTensor([T]), t, st, mu, prompt_lens=[Tp]).cpu()
            err = (got - want).abs()
            assert err.max().item() <= tol * scale and err.mean().item() <= 0.1 * tol * scale, (mode, err.max().item(), err.mean().item())
    finally:
        _lib.set_gemm_mode(_lib.GEMM_BF16X3)


def test_config4_vocoder_2580_frames_locality_vs_oracle(device, full):
    """BigVGAN on 2580 mel frames (30 s): an interior stretch of the waveform equals the CPU oracle run on a window of the mel
    around it (the vocoder's receptive field is finite: < 64 frames), both arithmetic modes."""
    from indextts_amd import _lib
    from indextts_amd.vocoder import BigVGAN
    from oracle import vocoder as ov
    cfg, wg, ws, wv = full
    voc = BigVGAN(wv, cfg.bigvgan)
    Tm, lo, hi, margin = 2580, 1200, 1392, 64
    mel = torch.from_numpy(weights.synth_mel("t/full/mel2580", 1, cfg.bigvgan.num_mels, Tm))
    torch.set_num_threads(16)
    wt = {k: torch.from_numpy(v) for k, v in wv.items()}
    with torch.no_grad():
        want = ov.bigvgan_forward(wt, cfg.bigvgan, mel[:, :, lo:hi])
    up = cfg.bigvgan.total_upsample
    a, b = margin * up, (hi - lo - margin) * up
    try:
        for mode, tol in ((_lib.GEMM_F32, 2e-4), (_lib.GEMM_BF16X3, 1.6e-3)):
            _lib.set_gemm_mode(mode)
            got = voc(mel.to(device)).cpu()
            assert got.shape == (1, 1, Tm * up)
            err = (got[..., lo * up + a: lo * up + b] - want[..., a:b]).abs().max().item()
            assert err <= tol, (mode, err)
    finally:
        _lib.set_gemm_mode(_lib.GEMM_BF16X3)


def test_config2_batch16_rows_equal_their_solo_runs(device, full):
    """configs[2] at full size (16 utterances x 128 text tokens x 512 codes, greedy): rows 3 and 11 of the batch produce the codes of
    their own B = 1 runs bit for bit, and mel within the north-star bound (L1 <= 1e-3) of the solo pipeline on the same noise rows."""
    from indextts_amd.infer_v2 import IndexTTS2
    cfg, wg, ws, wv = full
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=device)
    c = _cond(cfg)
    B, L, M, Tp = 16, 128, 512, 689
    text = torch.from_numpy(synth.integers("bench/text/rank0", (B, L), 2, cfg.gpt.number_text_tokens))
    Tg = int(M * cfg.code_to_frame)
    noise = torch.from_numpy(synth.uniform("bench/noise/rank0", (B, cfg.s2mel.in_channels, Tp + Tg), 1.7)).to(device)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wavs, mid = tts.synthesize_batch(text, c, max_mel_tokens=M, noise=noise, return_intermediates=True)
        assert mid["codes"].shape == (B, M) and all(n == M for n in mid["code_lens"])
        for b in (3, 11):
            w1, m1 = tts.synthesize_batch(text[b:b + 1], c, max_mel_tokens=M, noise=noise[b:b + 1], return_intermediates=True)
            assert torch.equal(m1["codes"][0], mid["codes"][b])
            l1 = (m1["mel"][0] - mid["mel"][b]).abs().mean().item()
            assert l1 <= 1e-3, l1
            assert (w1[0] - wavs[b]).abs().max().item() <= 32767 * 2e-3
    assert all(torch.isfinite(w).all() for w in wavs)


def test_prompt_conditioning_at_production_size_vs_oracle(device):
    """a8 at the real dimensions: conformer 6 x 512 (8 heads, 2048 units) + perceiver 32 x 1280 and the emotion pair (4 x 512 +
    1 latent x 1024) on a T = 750 prompt (15 s of w2v-bert frames): the K = 261 632 split-K input projection, 375 frames of
    rel-pos attention, against the CPU oracle."""
    from indextts_amd.cond import ConditioningEncoders
    from indextts_amd.config import GPTConfig
    from oracle import cond as oc
    cfg = GPTConfig()
    w = weights.synth_gpt_cond_weights(cfg, tag="t/full/cond")
    enc = ConditioningEncoders(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    T = 750
    x = torch.from_numpy(synth.uniform("t/full/cond/x", (1, T, 1024), 1.0))
    lens = torch.tensor([T])
    torch.set_num_threads(16)
    with torch.no_grad():
        ref_lat = oc.get_conditioning(tw, cfg, x, lens)
        ref_emo = oc.get_emovec(tw, cfg, x, lens)
    lat = enc.get_conditioning(x.transpose(1, 2), lens).cpu()
    ev = enc.get_emovec(x, lens).cpu()
    assert lat.shape == ref_lat.shape == (1, 32, cfg.model_dim) and ev.shape == ref_emo.shape == (1, cfg.model_dim)
    for got, want, name in ((lat, ref_lat, "conditioning latent"), (ev, ref_emo, "emotion vector")):
        err = (got - want).abs()
        scale = max(1.0, want.abs().max().item())
        assert torch.isfinite(got).all() and err.max().item() <= 2e-4 * scale and err.mean().item() <= 2e-5 * scale, (name, err.max().item(), err.mean().item(), scale)


def test_w2vbert_all_17_layers_at_15s_vs_oracle(device):
    """f1 at production size: the 17 w2v-bert-2.0 layers `hidden_states[17]` needs, hidden 1024 / 16 heads / ffn 4096, one 15 s
    prompt (T = 750 frames), against the CPU oracle (transformers' forward restated, oracle/semantic.py)."""
    from indextts_amd.config import W2VBertConfig
    from indextts_amd.semantic import SemanticModel
    from oracle import semantic as osem
    cfg = W2VBertConfig()
    assert cfg.num_layers == 17
    w = weights.synth_w2vbert_weights(cfg, tag="t/full/w2v")
    sm = SemanticModel(w, cfg, device=device)
    T = 750
    feats = torch.from_numpy(synth.uniform("t/full/w2v/feats", (1, T, cfg.input_dim), 1.5))
    mask = torch.ones(1, T, dtype=torch.long)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        want = osem.get_emb(tw, cfg, feats, mask).numpy()
    got = sm(feats, mask).cpu().numpy()
    scale = max(1.0, np.abs(want).max())
    err = np.abs(got - want)
    assert np.isfinite(got).all() and err.max() <= 1e-3 * scale and err.mean() <= 1e-4 * scale, (err.max(), err.mean(), scale)


def test_beam_search_at_full_size_16_utterances_vs_oracle(device, full):
    """f2 at production size: 16 utterances x 3 beams = 48 decode rows on the full-size GPT (the B * num_beams <= 64 limit of
    include/idxtts.h), deterministic beams and beam-sample with explicit Exp(1) draws, first 16 tokens against
    oracle.gpt.generate_beam; a raised stop bias lets hypotheses finish inside the window."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg, wg, ws, wv = full
    g = cfg.gpt
    wb = dict(wg)
    wb["mel_head.bias"] = synth.uniform("t/full/beam/bias", (g.number_mel_codes,), 0.5).astype(np.float32)
    wb["mel_head.bias"][g.stop_mel_token] = 8.5      # under sampling 5 of the 16 utterances finish inside the window (none deterministically)
    uv = UnifiedVoice(wb, g, device=device)
    c = _cond(cfg)
    B, L, NEW, NB = 16, 12, 16, 3
    text = torch.from_numpy(synth.integers("t/full/beam/text", (B, L), 2, g.number_text_tokens))
    tw = {k: torch.from_numpy(v) for k, v in wb.items()}
    conds = og.conds_latent(tw, g, c.spk_cond_latent.expand(B, -1, -1), c.emo_vec.expand(B, -1))
    gen = torch.Generator().manual_seed(7)
    noise = torch.stack([torch.empty(B, NB * g.number_mel_codes).exponential_(1, generator=gen) for _ in range(NEW)])
    torch.set_num_threads(16)
    for do_sample in (False, True):
        with torch.no_grad():
            want, trace = og.generate_beam(tw, g, conds, text, NEW, noise, num_beams=NB, do_sample=do_sample, return_trace=True)
        codes, _ = uv.inference_speech(c.spk_cond_latent.expand(B, -1, -1), text, emo_vec=c.emo_vec.expand(B, -1), max_generate_length=NEW,
                                       do_sample=do_sample, num_beams=NB, top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0,
                                       length_penalty=0.0, exp_noise=noise)
        got = codes.cpu().numpy()
        assert got.shape == tuple(want.shape), (got.shape, tuple(want.shape))
        rows = np.nonzero((got != want.numpy()).any(axis=1))[0]
        # exact-fp32 kernels vs torch CPU: summation order differs (scores agree to ~1e-4), so a near-tie between two candidates may fall
        # the other way once in a while.  At most one of the 16 utterances may differ, and only where the ORACLE's own decision was that
        # close: up to the first differing token, some step of that utterance must have had two neighbouring selection keys / candidate
        # scores within 2e-3 (sampling key: relative) -- otherwise the difference is a bug, not a tie.
        assert len(rows) <= 1, f"{len(rows)} of {B} utterances differ (do_sample={do_sample})"
        for b in rows:
            t = int(np.nonzero(got[b] != want.numpy()[b])[0][0])
            gaps = [float(tr[3][b]) for tr in trace[: t + 1]]
            assert min(gaps) <= 2e-3, f"utterance {b} differs from step {t} on, but the oracle's closest decision up to there had a gap of {min(gaps):.3e}"


def test_bf16_weights_and_kv_cache_16_utterances_vs_oracle(device, full):
    """configs[2]'s decode as bench.py runs it: the full-size GPT, 16 utterances (20 heads x 16 = 320 attention workgroups, no key
    split), bf16 weight streams AND the bf16 KV cache -- the first 16 greedy codes of every utterance against the CPU oracle on
    the read-back rounded model with the same key / value rounding (kv_round), ragged text lengths (left padding).  The rule of this
    mode (bench.py LOGIT_NOISE_BOUND): a key / value that two fp32 summation orders -- or the prefill's split-bf16
    GEMM and the oracle's fp32 one -- leave one ulp apart may round to the other bf16 neighbour, which moves a logit by up to 1.5e-2
    (measured 6e-3..7e-3); so a row either equals the oracle or leaves it at a step where the ORACLE's own margin between its token
    and the row's is below that bound, and at most two of the 16 rows do.  (The teacher-forced test below bounds every logit.)"""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg, wg, ws, wv = full
    g = cfg.gpt
    uv = UnifiedVoice(wg, g, device=device, weight_format="bf16", keep_effective=True)
    assert uv.kv_format == "bf16"
    tw = {k: torch.from_numpy(v) for k, v in uv.effective_state_dict.items()}
    c = _cond(cfg)
    B, L, NEW = 16, 40, 16
    text = torch.from_numpy(synth.integers("t/full/kv16/text", (B, L), 2, g.number_text_tokens))
    for b in range(B):
        text[b, L - (b % 5) * 3:] = g.stop_text_token
    lat, emo = c.spk_cond_latent.expand(B, -1, -1), c.emo_vec.expand(B, -1)
    torch.set_num_threads(16)
    with torch.no_grad():
        conds = og.conds_latent(tw, g, lat, emo)
        ref, ref_logits = og.generate_greedy(tw, g, conds, text, NEW, 10.0, return_logits=True, kv_round=True)
        fake = og.prepare_gpt_inputs(tw, g, conds, text)[0]
    from indextts_amd import _lib
    from test_gpt_gpu import _assert_equal_or_near_tie
    assert _lib.get_decode_geometry() is False
    try:
        for narrow in (False, True):      # idxtts_set_decode_geometry: the 1024-thread GEMVs and the 512-thread ones a serving loop selects
            _lib.set_decode_geometry(narrow)
            assert _lib.get_decode_geometry() is narrow
            codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, do_sample=False, num_beams=1, repetition_penalty=10.0)
            got = codes.cpu().numpy()
            assert got.shape == tuple(ref.shape), narrow
            assert _assert_equal_or_near_tie(got, ref.numpy(), ref_logits, fake, 1.5e-2) <= 2, narrow
    finally:
        _lib.set_decode_geometry(False)


def _parity_setup(full, device, kv_format):
    from indextts_amd.gpt import UnifiedVoice
    cfg, wg, ws, wv = full
    g = cfg.gpt
    uv = UnifiedVoice(wg, g, device=device, weight_format="bf16", keep_effective=True, kv_format=kv_format)
    tw = {k: torch.from_numpy(v) for k, v in uv.effective_state_dict.items()}
    c = _cond(cfg)
    B, L = 6, 32      # more than 4 rows: the plane-GEMV decode step, as in the bench's 16-row batches
    text = torch.from_numpy(synth.integers("bench/text/rank0", (16, 128), 2, g.number_text_tokens))[:B, :L].clone()      # bench.py's CPU-leg prefixes
    return uv, tw, g, c.spk_cond_latent.expand(B, -1, -1).contiguous(), c.emo_vec.expand(B, -1).contiguous(), text


# What the two storage modes of the decode may add to a logit against the fp32 CPU oracle running the same (rounded) model, full size
# (logit std 3.1).  Measured on MI355X, 8 utterances x 96 steps: fp32 KV cache max 1.7e-5 (summation order only), bf16 KV cache max
# 5.8e-3, mean 7.8e-4 (a key / value one ulp apart between two summation orders may round to the other bf16 neighbour: 2^-9 relative).
# bench.py reports the same quantities on its CPU-leg utterances in every run.
LOGIT_BOUND = {"f32": 1e-4, "bf16": 1.5e-2}


@pytest.mark.parametrize("kv_format", ["bf16", "f32"])
def test_decode_teacher_forced_logit_bound_and_every_code_flip_is_a_near_tie(device, full, kv_format):
    """The bench's decode mode (bf16 weight streams; bf16 or fp32 KV cache), full-size GPT, 6 utterances x 96 steps TEACHER-FORCED on the
    oracle's codes (oracle/parity.py): every logit within LOGIT_BOUND of the oracle's, every step whose argmax differs from the oracle's
    token is a near-tie of the ORACLE (margin below twice the bound), and the free-running decode leaves the oracle's sequence exactly
    at the first such step of an utterance -- nowhere else."""
    from oracle import parity
    uv, tw, g, lat, emo, text = _parity_setup(full, device, kv_format)
    torch.set_num_threads(16)
    r = parity.decode_parity(uv, tw, g, lat, emo, text, 96)
    bound = LOGIT_BOUND[kv_format]
    print(f"[parity {kv_format}] max|dlogit| {r['max_abs_logit_diff']:.3e} (mean {r['mean_abs_logit_diff']:.3e}, logit std {r['logit_std']:.2f}), "
          f"teacher-forced match rate {r['codes_match_rate_teacher_forced']:.4f}, mismatches {r['mismatching_steps']}, "
          f"oracle top-2 margin median {r['oracle_top2_margin_median']:.3e} min {r['oracle_top2_margin_min']:.3e}, free-running first differences {r['first_difference_step_free_running']}")
    assert r["steps"] == 96 and r["utterances"] == 6
    assert r["max_abs_logit_diff"] <= bound, r["max_abs_logit_diff"]
    for t in r["mismatching_steps"]:
        assert 0.0 <= t["oracle_margin_to_hip_token"] <= 2 * bound, t
    assert r["codes_match_rate_teacher_forced"] >= 0.97
    assert r["first_difference_step_free_running"] == r["first_mismatch_step_teacher_forced"]
