"""GPU parity: the s2mel stage (gpt_layer + vq2emb + length regulator, CFM/DiT/WaveNet) through the C ABI against the
reference's golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import S2MelConfig

pytestmark = pytest.mark.gpu


def _tiny(golden_dir, device):
    from indextts_amd.s2mel import S2Mel
    g = np.load(os.path.join(golden_dir, "s2mel.npz"))
    cfg = S2MelConfig.tiny()
    w = weights.synth_s2mel_weights(cfg, tag="golden/s2mel")
    return g, cfg, w, S2Mel(w, cfg, device=device)


def test_constant_tables_match_oracle():
    from indextts_amd.s2mel import rope_cache, timestep_tables
    from oracle import s2mel as osm
    assert torch.equal(rope_cache(100, 64), osm.rope_cache(100, 64))
    emb, dt = timestep_tables(4)
    t = torch.tensor([0.0, 0.25, 0.5, 0.75])
    assert torch.allclose(emb, osm.timestep_embedding(t), atol=1e-6)


def test_condition_path_vs_reference_golden(device, golden_dir):
    """gpt_layer + vq2emb + length regulator for the two golden utterances, run as ONE ragged batch: per-utterance
    GroupNorm statistics, interpolation and zero padding must reproduce the reference's B=1 results."""
    from oracle import s2mel as osm
    g, cfg, w, sm = _tiny(golden_dir, device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    Ms = {"a": 9, "b": 20}
    lat = {t: torch.from_numpy(synth.uniform(f"t/s2mel/lat_{t}", (1, M, cfg.gpt_dim), 1.0)) for t, M in Ms.items()}
    codes = {t: torch.from_numpy(synth.integers(f"t/s2mel/codes_{t}", (1, M), 0, cfg.codebook_size)) for t, M in Ms.items()}
    # oracle per utterance (its pieces are pinned to the reference by tests/test_oracle_s2mel.py)
    ref = {}
    for t, M in Ms.items():
        S = osm.vq2emb(tw, codes[t]) + osm.gpt_layer(tw, lat[t])
        ref[t] = osm.length_regulator(tw, cfg, S, (torch.LongTensor([M]) * 1.72).long())
    Mmax = 20
    latb = torch.zeros(2, Mmax, cfg.gpt_dim)
    cb = torch.zeros(2, Mmax, dtype=torch.long)
    latb[0, :9], latb[1] = lat["a"][0], lat["b"][0]
    cb[0, :9], cb[1] = codes["a"][0], codes["b"][0]
    cond, tl = sm.prepare_condition(latb.to(device), cb.to(device), torch.tensor([9, 20]))
    assert tl.tolist() == [15, 34] and cond.shape == (2, 34, cfg.lr_channels)
    got = cond.cpu()
    assert (got[0, :15] - ref["a"][0]).abs().max().item() <= 3e-5
    assert (got[1] - ref["b"][0]).abs().max().item() <= 3e-5
    assert (got[0, 15:] == 0).all()
    # and directly against what the reference's own modules produced for these two utterances (infer_v2.py:835-849 composed)
    assert (got[0, :15] - torch.from_numpy(g["cond_a"])[0]).abs().max().item() <= 5e-5
    assert (got[1] - torch.from_numpy(g["cond_b"])[0]).abs().max().item() <= 5e-5


def test_length_regulator_vs_reference_golden(device, golden_dir):
    """`length_regulator(S, ylens)` alone (the S_ref -> prompt_condition call, infer_v2.py:649-652) against the reference module's
    own outputs for two lengths."""
    g, cfg, w, sm = _tiny(golden_dir, device)
    for tag, M in (("a", 9), ("b", 20)):
        S = torch.from_numpy(synth.uniform(f"golden/s2mel/S_{tag}", (1, M, cfg.lr_in_channels), 1.0))
        ylens = (torch.LongTensor([M]) * 1.72).long()
        cond, _ = sm.length_regulator(S, ylens, n_quantizers=3, f0=None)
        np.testing.assert_allclose(cond.cpu().numpy(), g[f"lr_{tag}"], rtol=0, atol=5e-5)


def test_gpt_layer_and_vq2emb_vs_reference_golden(device, golden_dir):
    """`gpt_layer(latent)` and `vq2emb(codes)` fixtures of the reference: the HIP condition path is linear in each up to the
    length regulator, so they are checked through the one place their sum is visible -- content_in_proj has no nonlinearity
    before the first conv, hence compare prepare_condition against the oracle fed with the REFERENCE's two tensors."""
    from oracle import s2mel as osm
    g, cfg, w, sm = _tiny(golden_dir, device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    lat = torch.from_numpy(synth.uniform("golden/s2mel/latent", (2, 9, cfg.gpt_dim), 1.0))
    codes = torch.from_numpy(synth.integers("golden/s2mel/codes", (2, 9), 0, cfg.codebook_size))
    cond, tl = sm.prepare_condition(lat.to(device), codes.to(device), torch.tensor([9, 9]))
    S_ref = torch.from_numpy(g["vq2emb"]) + torch.from_numpy(g["gpt_layer"])          # the reference's own outputs
    for b in range(2):
        want = osm.length_regulator(tw, cfg, S_ref[b:b + 1], (torch.LongTensor([9]) * 1.72).long())
        assert (cond[b].cpu() - want[0]).abs().max().item() <= 5e-5


def test_estimator_vs_reference_golden(device, golden_dir):
    """One DiT.forward (diffusion_transformer.py:186-257) on two rows with different x_lens, directly against the reference's output."""
    g, cfg, w, sm = _tiny(golden_dir, device)
    T = 37
    x = torch.from_numpy(synth.uniform("golden/s2mel/dit/x", (2, cfg.in_channels, T), 1.0))
    px = torch.from_numpy(synth.uniform("golden/s2mel/dit/prompt", (2, cfg.in_channels, T), 1.0))
    px[..., 12:] = 0
    st = torch.from_numpy(synth.uniform("golden/s2mel/dit/style", (2, cfg.style_dim), 1.0))
    mu = torch.from_numpy(synth.uniform("golden/s2mel/dit/mu", (2, T, cfg.content_dim), 1.0))
    from indextts_amd import _lib
    want = torch.from_numpy(g["dit"])
    alone = torch.from_numpy(g["dit_row1_alone"])
    lens = torch.LongTensor([T, T - 6])
    halo = cfg.wn_layers * (cfg.wn_kernel // 2)          # frames of row 1 that see its end through the WaveNet convolutions
    try:
        for mode, tol in ((_lib.GEMM_F32, 1e-4), (_lib.GEMM_BF16X3, 4e-4)):
            _lib.set_gemm_mode(mode)
            got = sm.estimator(x, px, lens, torch.tensor([0.35, 0.35]), st, mu, prompt_lens=[12, 12]).cpu()
            scale = max(1.0, want.abs().max().item())
            assert (got[0] - want[0]).abs().max().item() <= tol * scale, mode                      # full-length row: the padded-batch fixture
            # the short row equals the reference's call on that row ALONE (infer_v2 only ever runs B = 1, x_lens = T) on every
            # frame; the padded-batch fixture agrees with that away from the row's end (there the reference's batched call
            # convolves over the zero padding instead of reflecting, a case infer_v2 never produces)
            assert (got[1, :, :T - 6] - alone[0]).abs().max().item() <= tol * scale, mode
            assert (got[1, :, :T - 6 - halo] - want[1, :, :T - 6 - halo]).abs().max().item() <= tol * scale, mode
    finally:
        _lib.set_gemm_mode(_lib.GEMM_BF16X3)


def test_cfm_vs_reference_golden(device, golden_dir):
    g, cfg, w, sm = _tiny(golden_dir, device)
    Tp, Tg = 11, 23
    T = Tp + Tg
    z = torch.from_numpy(synth.uniform("golden/s2mel/cfm/z", (1, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("golden/s2mel/cfm/mu", (1, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("golden/s2mel/cfm/prompt", (1, cfg.in_channels, Tp), 1.0))
    st = torch.from_numpy(synth.uniform("golden/s2mel/cfm/style", (1, cfg.style_dim), 1.0))
    out = sm.cfm_inference(mu, torch.LongTensor([T]), prompt, st, None, 3, inference_cfg_rate=0.7, z=z).cpu()
    err = (out - torch.from_numpy(g["cfm"])).abs()
    assert err.max().item() <= 2e-4, err.max().item()
    assert err.mean().item() <= 2e-5                      # mel L1 (north_star bound: 1e-3)
    assert (out[..., :Tp] == 0).all()


def test_cfm_ragged_batch_vs_oracle(device, golden_dir):
    """Three utterances of different total and prompt lengths in one padded batch == the oracle run row by row
    (key-padding mask, per-sequence reflect padding in the WaveNet, row masks)."""
    from oracle import s2mel as osm
    g, cfg, w, sm = _tiny(golden_dir, device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    lens, plens = [41, 29, 36], [10, 7, 12]
    B, T, Tpm = 3, max(lens), max(plens)
    z = torch.from_numpy(synth.uniform("t/s2mel/rag/z", (B, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/rag/mu", (B, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/rag/prompt", (B, cfg.in_channels, Tpm), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/rag/style", (B, cfg.style_dim), 1.0))
    for b in range(B):
        mu[b, lens[b]:] = 0
    out = sm.cfm_inference(mu, torch.LongTensor(lens), prompt, st, None, 2, inference_cfg_rate=0.7, z=z,
                           prompt_lens=torch.LongTensor(plens)).cpu()
    for b in range(B):
        Lb, Pb = lens[b], plens[b]
        ref = osm.cfm_inference(tw, cfg, mu[b:b + 1, :Lb], torch.LongTensor([Lb]), prompt[b:b + 1, :, :Pb], st[b:b + 1],
                                z[b:b + 1, :, :Lb], 2, 0.7)
        assert (out[b, :, :Lb] - ref[0]).abs().max().item() <= 2e-4, b


def test_cfm_full_width_smoke(device):
    """Full IndexTTS-2 s2mel widths (DiT 512x13, WaveNet 512x8) on a short sequence vs the CPU oracle."""
    from indextts_amd.s2mel import S2Mel
    from oracle import s2mel as osm
    cfg = S2MelConfig()
    w = weights.synth_s2mel_weights(cfg, tag="t/s2mel/full")
    sm = S2Mel(w, cfg, device=device, max_frames=512)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, Tp, Tg = 1, 24, 40
    T = Tp + Tg
    z = torch.from_numpy(synth.uniform("t/s2mel/full/z", (B, 80, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/full/mu", (B, T, 512), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/full/prompt", (B, 80, Tp), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/full/style", (B, 192), 1.0))
    out = sm.cfm_inference(mu, torch.LongTensor([T]), prompt, st, None, 2, inference_cfg_rate=0.7, z=z).cpu()
    ref = osm.cfm_inference(tw, cfg, mu, torch.LongTensor([T]), prompt, st, z, 2, 0.7)
    err = (out - ref).abs()
    assert err.max().item() <= 5e-4 * max(1.0, ref.abs().max().item()), err.max().item()
    assert err.mean().item() <= 1e-4


def test_cfm_split_bf16_mode_vs_oracle(device, golden_dir):
    """Sequences long enough (2B*T >= 256 rows) to take the split-bf16 GEMM path: mel error stays far inside the
    north-star bound (L1 <= 1e-3), and the exact-fp32 mode is available and tighter."""
    from indextts_amd import _lib
    from oracle import s2mel as osm
    g, cfg, w, sm = _tiny(golden_dir, device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, Tp, T = 1, 40, 170
    z = torch.from_numpy(synth.uniform("t/s2mel/b16/z", (B, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/b16/mu", (B, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/b16/prompt", (B, cfg.in_channels, Tp), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/b16/style", (B, cfg.style_dim), 1.0))
    ref = osm.cfm_inference(tw, cfg, mu, torch.LongTensor([T]), prompt, st, z, 3, 0.7)
    res = {}
    try:
        for name, mode in (("f32", _lib.GEMM_F32), ("bf16x3", _lib.GEMM_BF16X3)):
            _lib.set_gemm_mode(mode)
            out = sm.cfm_inference(mu, torch.LongTensor([T]), prompt, st, None, 3, inference_cfg_rate=0.7, z=z).cpu()
            res[name] = (out - ref).abs()
    finally:
        _lib.set_gemm_mode(_lib.GEMM_BF16X3)
    assert res["f32"].max().item() <= 3e-4 and res["f32"].mean().item() <= 2e-5
    assert res["bf16x3"].mean().item() <= 1e-4 and res["bf16x3"].max().item() <= 3e-3      # bound: mel L1 <= 1e-3
    assert res["bf16x3"].mean().item() > 0


def test_cfm_long_batch_dma_gemm_chain_vs_oracle(device):
    """2B*T >= 4096 rows and every projection >= 192 columns: the DiT / WaveNet GEMMs run on the LDS-DMA split-bf16 kernel,
    producers hand their outputs over as bf16 hi/lo planes (adaLN norm, attention, SwiGLU and gate epilogues), the rotary
    embedding rides in the qkv epilogue and attention runs in its split-bf16 form -- against the fp32 CPU oracle, ragged."""
    import dataclasses
    from indextts_amd.s2mel import S2Mel
    from oracle import s2mel as osm
    cfg = dataclasses.replace(S2MelConfig.tiny(), hidden_dim=512, num_heads=8, depth=3, wn_hidden=512, wn_layers=2, block_size=1024)
    w = weights.synth_s2mel_weights(cfg, tag="t/s2mel/chain")
    sm = S2Mel(w, cfg, device=device, max_frames=1024)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    lens, plens = [720, 655, 701], [60, 33, 48]
    B, T, Tpm = 3, max(lens), max(plens)
    z = torch.from_numpy(synth.uniform("t/s2mel/chain/z", (B, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/chain/mu", (B, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/chain/prompt", (B, cfg.in_channels, Tpm), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/chain/style", (B, cfg.style_dim), 1.0))
    for b in range(B):
        mu[b, lens[b]:] = 0
    out = sm.cfm_inference(mu, torch.LongTensor(lens), prompt, st, None, 2, inference_cfg_rate=0.7, z=z,
                           prompt_lens=torch.LongTensor(plens)).cpu()
    for b in range(B):
        Lb, Pb = lens[b], plens[b]
        ref = osm.cfm_inference(tw, cfg, mu[b:b + 1, :Lb], torch.LongTensor([Lb]), prompt[b:b + 1, :, :Pb], st[b:b + 1],
                                z[b:b + 1, :, :Lb], 2, 0.7)
        err = (out[b, :, :Lb] - ref[0]).abs()
        assert err.mean().item() <= 1e-4 and err.max().item() <= 3e-3, (b, err.mean().item(), err.max().item())


@pytest.mark.parametrize("lens,plens", [([720, 655, 701], [300, 260, 281]), ([1100, 1100], [400, 400])])
def test_cfm_long_prompts_tail_only_wavenet_vs_oracle(device, lens, plens):
    """Prompts longer than 64 frames + the WaveNet's context: the solver evaluates the post-transformer part (long skip, WaveNet,
    final layer) only from frame min(prompt_len) - halo on, on compacted rows -- the Euler step discards the prompt frames anyway
    (flow_matching.py:113).  Ragged batch vs the per-utterance CPU oracle; the second case keeps the compacted rows on the LDS-DMA
    GEMM chain (>= 4096 rows), the first drops to the fp32-row kernels for the tail."""
    import dataclasses
    from indextts_amd.s2mel import S2Mel
    from oracle import s2mel as osm
    cfg = dataclasses.replace(S2MelConfig.tiny(), hidden_dim=512, num_heads=8, depth=3, wn_hidden=512, wn_layers=2, block_size=2048)
    w = weights.synth_s2mel_weights(cfg, tag="t/s2mel/tail")
    sm = S2Mel(w, cfg, device=device, max_frames=2048)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, T, Tpm = len(lens), max(lens), max(plens)
    z = torch.from_numpy(synth.uniform("t/s2mel/tail/z", (B, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/tail/mu", (B, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/tail/prompt", (B, cfg.in_channels, Tpm), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/tail/style", (B, cfg.style_dim), 1.0))
    for b in range(B):
        mu[b, lens[b]:] = 0
    out = sm.cfm_inference(mu, torch.LongTensor(lens), prompt, st, None, 2, inference_cfg_rate=0.7, z=z,
                           prompt_lens=torch.LongTensor(plens)).cpu()
    torch.set_num_threads(16)
    for b in range(B):
        Lb, Pb = lens[b], plens[b]
        ref = osm.cfm_inference(tw, cfg, mu[b:b + 1, :Lb], torch.LongTensor([Lb]), prompt[b:b + 1, :, :Pb], st[b:b + 1],
                                z[b:b + 1, :, :Lb], 2, 0.7)
        err = (out[b, :, :Lb] - ref[0]).abs()
        assert err.max().item() <= 3e-3 and err.mean().item() <= 1e-4, (b, err.max().item(), err.mean().item())
        assert (out[b, :, :Pb] == 0).all()


def test_cfm_halves_on_two_streams_equal_the_stacked_batch(device):
    """idxtts_s2mel_set_overlap(1): the conditional and unconditional halves of every CFM step on two streams instead of one stacked
    2B batch (s2mel.hip::dit_eval_halves).  Rows of a batch are computed independently, so the mel must not change by one bit
    (both forms keep every GEMM on the LDS-DMA kernel: B*T >= 256 rows)."""
    import dataclasses
    from indextts_amd import _lib
    from indextts_amd.s2mel import S2Mel
    cfg = dataclasses.replace(S2MelConfig.tiny(), hidden_dim=512, num_heads=8, depth=3, wn_hidden=512, wn_layers=2, block_size=1024)
    w = weights.synth_s2mel_weights(cfg, tag="t/s2mel/chain")
    sm = S2Mel(w, cfg, device=device, max_frames=1024)
    lens, plens = [720, 655, 701], [60, 33, 48]
    B, T, Tpm = 3, max(lens), max(plens)
    z = torch.from_numpy(synth.uniform("t/s2mel/chain/z", (B, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("t/s2mel/chain/mu", (B, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("t/s2mel/chain/prompt", (B, cfg.in_channels, Tpm), 1.0))
    st = torch.from_numpy(synth.uniform("t/s2mel/chain/style", (B, cfg.style_dim), 1.0))
    for b in range(B):
        mu[b, lens[b]:] = 0
    assert _lib.get_s2mel_overlap() == 0              # the shipped default
    outs = {}
    try:
        for mode in (0, 1, 0):
            _lib.set_s2mel_overlap(mode)
            assert _lib.get_s2mel_overlap() == mode
            o = sm.cfm_inference(mu, torch.LongTensor(lens), prompt, st, None, 3, inference_cfg_rate=0.7, z=z,
                                 prompt_lens=torch.LongTensor(plens)).cpu()
            assert torch.isfinite(o).all()
            outs.setdefault(mode, []).append(o)
    finally:
        _lib.set_s2mel_overlap(0)
    assert torch.equal(outs[0][0], outs[0][1])
    assert torch.equal(outs[0][0], outs[1][0])
