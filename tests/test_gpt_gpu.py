"""GPU parity: token-major building blocks (GEMM, attention, LayerNorm) and the GPT stage (greedy decode,
latent pass) through the C ABI, against torch fp64/fp32 CPU references, the CPU oracle and the HF golden vectors."""
import ctypes
import math
import os
from ctypes import c_void_p

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from indextts_amd import _lib, synth, weights
from indextts_amd.config import GPTConfig

pytestmark = pytest.mark.gpu


def _linear(device, w, b, x, act=0, res=None, kn=False, bf16x3=0):
    lib = _lib.load()
    N, K = (w.shape[1], w.shape[0]) if kn else w.shape
    h = c_void_p()
    wd = w.contiguous()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(wd), _lib.ptr(b), N, K, int(kn), ctypes.byref(h)))
    xd = x.to(device).contiguous()
    M = xd.shape[0]
    No = N // 2 if act == 3 else N
    y = torch.empty(M, No, device=device)
    rd = None if res is None else res.to(device).contiguous()
    _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(xd), xd.shape[1], _lib.ptr(y), No, _lib.ptr(rd), No, M, act, bf16x3, _lib.current_stream()))
    out = y.cpu()
    lib.idxtts_linear_destroy(h)
    return out


@pytest.mark.parametrize("shape", [(1, 16, 4), (7, 40, 36), (128, 128, 128), (300, 1280, 256), (165, 3840, 1280), (1000, 512, 864),
                                   (129, 80, 512), (64, 8194, 128)])
def test_gemm_tn_vs_torch(device, shape):
    M, N, K = shape
    x = torch.from_numpy(synth.uniform(f"t/gemm/x/{shape}", (M, K), 1.0))
    w = torch.from_numpy(synth.fan_in_uniform(f"t/gemm/w/{shape}", (N, K), K))
    b = torch.from_numpy(synth.uniform(f"t/gemm/b/{shape}", (N,), 0.2))
    ref = (x.double() @ w.double().t() + b.double()).float()
    y = _linear(device, w, b, x)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # HF Conv1D layout + gelu_new + residual
    r = torch.from_numpy(synth.uniform(f"t/gemm/r/{shape}", (M, N), 1.0))
    pre = x.double() @ w.double().t() + b.double()
    gelu = 0.5 * pre * (1 + torch.tanh(math.sqrt(2 / math.pi) * (pre + 0.044715 * pre ** 3)))
    y2 = _linear(device, w.t().contiguous(), b, x, act=1, res=r, kn=True)
    assert (y2 - (gelu + r.double()).float()).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,K", [(1280, 1280), (3840, 1280), (1280, 5120), (512, 864), (256, 96)])
def test_gemm_tn_rows_do_not_depend_on_the_row_count(device, N, K):
    """A launch with few output tiles puts one K group per workgroup and combines them in the kernel (last arriver); a launch with many tiles
    runs the whole K loop in one workgroup.  Both add the same K groups in the same order: rows computed alone (split) equal, bit for bit,
    the same rows inside a 2000-row call (unsplit) -- what keeps a B = 1 prefill identical to its rows in a batched one."""
    M = 2000
    x = torch.from_numpy(synth.uniform(f"t/gemm/inv/x/{N}/{K}", (M, K), 1.0))
    w = torch.from_numpy(synth.fan_in_uniform(f"t/gemm/inv/w/{N}/{K}", (N, K), K))
    b = torch.from_numpy(synth.uniform(f"t/gemm/inv/b/{N}/{K}", (N,), 0.2))
    r = torch.from_numpy(synth.uniform(f"t/gemm/inv/r/{N}/{K}", (M, N), 1.0))
    big = _linear(device, w, b, x, act=1, res=r)
    for lo, hi in ((0, 77), (130, 258), (1990, 2000)):
        small = _linear(device, w, b, x[lo:hi].contiguous(), act=1, res=r[lo:hi].contiguous())
        assert torch.equal(small, big[lo:hi]), (lo, hi)
    ref = x.double() @ w.double().t() + b.double()
    gelu = 0.5 * ref * (1 + torch.tanh(math.sqrt(2 / math.pi) * (ref + 0.044715 * ref ** 3)))
    assert (big - (gelu + r.double()).float()).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(256, 128, 32), (300, 1536, 512), (1000, 512, 864), (129, 80, 512), (640, 512, 1536), (513, 8194, 128),
                                   (4100, 1536, 512), (5003, 80, 512), (4096, 320, 96),     # >= 4096 rows: the 256-row tile kernel
                                   (16640, 1024, 512), (16500, 1280, 256)])                # 260 / 325 tiles: K-split tail tiles
def test_gemm_split_bf16_vs_torch(device, shape):
    """Split-bf16 GEMM (3 bf16 MFMAs per product): relative error ~2^-16 per product, i.e. ~1e-5 of the row scale."""
    M, N, K = shape
    x = torch.from_numpy(synth.uniform(f"t/gemm16/x/{shape}", (M, K), 1.0))
    w = torch.from_numpy(synth.fan_in_uniform(f"t/gemm16/w/{shape}", (N, K), K))
    b = torch.from_numpy(synth.uniform(f"t/gemm16/b/{shape}", (N,), 0.2))
    ref = (x.double() @ w.double().t() + b.double()).float()
    y = _linear(device, w, b, x, bf16x3=1)
    err = (y - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    assert (y - ref).abs().mean().item() <= 1e-5
    # and it really is more accurate than plain bf16 would be (sanity: plain bf16 error would be ~4e-3)
    assert err < 1e-3


def test_gemm_swiglu_and_silu(device):
    M, Hd, K = 200, 192, 64      # hidden 192 -> N = 384 packed as [32 w1 | 32 w3] blocks
    x = torch.from_numpy(synth.uniform("t/swiglu/x", (M, K), 1.0))
    w1 = torch.from_numpy(synth.fan_in_uniform("t/swiglu/w1", (Hd, K), K, 2.0))
    w3 = torch.from_numpy(synth.fan_in_uniform("t/swiglu/w3", (Hd, K), K, 2.0))
    packed = torch.stack([w1.view(Hd // 32, 32, K), w3.view(Hd // 32, 32, K)], dim=1).reshape(2 * Hd, K)
    y = _linear(device, packed, None, x, act=3)
    ref = (F.silu(x.double() @ w1.double().t()) * (x.double() @ w3.double().t())).float()
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    ys = _linear(device, w1, None, x, act=2)
    assert (ys - F.silu(x.double() @ w1.double().t()).float()).abs().max().item() <= 2e-5


@pytest.mark.parametrize("cfg", [(2, 2, 37, True, True), (1, 4, 165, True, False), (3, 2, 300, False, True), (2, 8, 129, False, False),
                                 (1, 2, 1, True, False), (1, 1, 700, True, True)])
def test_attention_vs_torch(device, cfg):
    B, H, S, causal, ragged = cfg
    d = H * 64
    qkv = torch.from_numpy(synth.uniform(f"t/attn/qkv/{cfg}", (B, S, 3 * d), 1.5))
    kstart = torch.zeros(B, dtype=torch.int32)
    kend = torch.full((B,), S, dtype=torch.int32)
    if ragged:
        for b in range(B):
            if causal:
                kstart[b] = (b * 5 + 3) % max(1, S // 2)       # left padding (GPT prompts)
            else:
                kend[b] = S - (b * 7) % max(1, S // 2)         # right padding (DiT x_lens)
    q, k, v = (t.view(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(d, dim=2))
    pos = torch.arange(S)
    allowed = (pos[None, :] >= kstart[:, None]) & (pos[None, :] < kend[:, None])          # [B,S] keys
    allowed = allowed[:, None, None, :].expand(B, 1, S, S)
    if causal:
        allowed = allowed & (pos[None, :] <= pos[:, None])[None, None]
    scores = (q @ k.transpose(-1, -2)) / 8.0
    scores = scores.masked_fill(~allowed, float("-inf"))
    att = torch.softmax(scores, -1)
    att = torch.nan_to_num(att, nan=0.0)            # rows with no visible key -> zeros (kernel convention)
    ref = (att @ v).transpose(1, 2).reshape(B, S, d).float()
    lib = _lib.load()
    qd = qkv.to(device).contiguous()
    o = torch.empty(B, S, d, device=device)
    ks, ke = kstart.to(device), kend.to(device)
    base = qd.data_ptr()
    valid = allowed.any(-1)[:, 0, :]                 # [B,S] query rows with at least one key
    # exact-fp32 MFMA form, then the split-bf16 form (3 bf16 MFMAs per product: ~2^-16 relative per product)
    for fn, tol in ((lib.idxtts_attention_fwd, 2e-5), (lib.idxtts_attention_bf16x3_fwd, 1e-4)):
        o.zero_()
        _lib.check(fn(c_void_p(base), c_void_p(base + 4 * d), c_void_p(base + 8 * d), _lib.ptr(o), S * 3 * d, 3 * d,
                      S * 3 * d, 3 * d, S * d, d, B, H, S, S, int(causal), _lib.ptr(ks), _lib.ptr(ke), 0.125, _lib.current_stream()))
        got = o.cpu()
        err = ((got - ref).abs() * valid[:, :, None]).max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (fn.__name__, err)
        assert (got[~valid].abs().max().item() if (~valid).any() else 0.0) == 0.0


@pytest.mark.parametrize("B,H,S,left,right", [(2, 3, 300, 64, 8), (1, 2, 77, 64, 8), (2, 16, 750, 64, 8), (1, 1, 40, 5, 90), (2, 2, 129, 0, 0)])
def test_attention_relative_key_vs_torch(device, B, H, S, left, right):
    """The w2v-bert self-attention of the prompt block (HF Wav2Vec2BertSelfAttention, position_embeddings_type = "relative_key") on the MFMA
    flash kernels: scores += q . rel_key[clamp(j - i, -left, right) + left] / sqrt(dk), right-padded keys masked; exact-fp32 and
    split-bf16 forms against float64 torch."""
    d = H * 64
    qkv = torch.from_numpy(synth.uniform(f"t/attn/rel/qkv/{B}/{H}/{S}", (B, S, 3 * d), 1.0))
    rel = torch.from_numpy(synth.uniform(f"t/attn/rel/tab/{left}/{right}", (left + right + 1, 64), 1.0))
    kend = torch.tensor([S - 7 * b for b in range(B)], dtype=torch.int32)
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(d, dim=-1)]
    pos = torch.arange(S)
    dist = (pos[None, :] - pos[:, None]).clamp(-left, right) + left                       # [i, j] -> table row
    scores = (q @ k.transpose(-1, -2) + torch.einsum("bhid,ijd->bhij", q, rel.double()[dist])) / 8.0
    scores = scores.masked_fill(~(pos[None, :] < kend[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(scores, -1) @ v).transpose(1, 2).reshape(B, S, d).float()
    lib = _lib.load()
    qd, rd, ke = qkv.to(device).contiguous(), rel.to(device).contiguous(), kend.to(device)
    o = torch.empty(B, S, d, device=device)
    base = qd.data_ptr()
    for split, tol in ((0, 2e-5), (1, 1e-4)):
        o.zero_()
        _lib.check(lib.idxtts_attention_relkey_fwd(c_void_p(base), c_void_p(base + 4 * d), c_void_p(base + 8 * d), _lib.ptr(o), S * 3 * d, 3 * d,
                                                   S * d, d, B, H, S, _lib.ptr(ke), 0.125, _lib.ptr(rd), left, right, split, _lib.current_stream()))
        err = (o.cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (split, err)


def test_layernorm_vs_torch(device):
    for (M, d) in [(5, 128), (33, 1280), (4, 5120), (3, 512)]:
        x = torch.from_numpy(synth.uniform(f"t/ln/x/{M}/{d}", (M, d), 3.0, 0.7))
        g = torch.from_numpy(synth.uniform(f"t/ln/g/{d}", (d,), 0.5, 1.0))
        b = torch.from_numpy(synth.uniform(f"t/ln/b/{d}", (d,), 0.5))
        y = torch.empty(M, d, device=device)
        xd, gd, bd = x.to(device), g.to(device), b.to(device)     # keep the device copies alive across the launch
        _lib.check(_lib.load().idxtts_layernorm_fwd(_lib.ptr(xd), _lib.ptr(y), _lib.ptr(gd), _lib.ptr(bd),
                                                    M, d, 1e-5, _lib.current_stream()))
        ref = F.layer_norm(x.double(), (d,), g.double(), b.double(), 1e-5).float()
        assert (y.cpu() - ref).abs().max().item() <= 1e-5


# ------------------------------------------------------------------------------------------------
def _tiny(golden_dir, device):
    from indextts_amd.gpt import UnifiedVoice
    g = np.load(os.path.join(golden_dir, "gpt.npz"))
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="golden/gpt")
    return g, cfg, w, UnifiedVoice(w, cfg, device=device)


def test_greedy_codes_bit_exact_vs_hf_golden(device, golden_dir):
    g, cfg, w, uv = _tiny(golden_dir, device)
    B, L = g["greedy_text"].shape
    NEW = g["greedy_codes"].shape[1]
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5)).to(device)
    text = torch.from_numpy(g["greedy_text"])
    fake, emb, mask = uv.prepare_gpt_inputs(conds, text)
    for graph in (False, True):
        out = uv.generate(fake, max_new_tokens=NEW, stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb,
                          repetition_penalty=10.0, use_graph=graph)
        codes = out[:, fake.shape[1]:].cpu().numpy()
        assert np.array_equal(codes, g["greedy_codes"]), f"graph={graph}"       # token indices: bit-exact
    out, logits = uv.generate(fake, max_new_tokens=NEW, stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb,
                              repetition_penalty=10.0, return_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["greedy_logits"], rtol=0, atol=5e-4)


def test_prepare_inputs_matches_oracle(device, golden_dir):
    from oracle import gpt as og
    g, cfg, w, uv = _tiny(golden_dir, device)
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (3, cfg.cond_latents + 2, cfg.model_dim), 0.5))
    text = torch.from_numpy(g["greedy_text"])
    fake, emb, mask = uv.prepare_gpt_inputs(conds.to(device), text)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    f2, e2, m2 = og.prepare_gpt_inputs(tw, cfg, conds, text)
    assert torch.equal(fake, f2) and torch.equal(mask, m2)
    assert torch.equal(emb.cpu(), e2)        # pure gathers + one add: bit-exact


def test_latent_pass_vs_golden(device, golden_dir):
    g, cfg, w, uv = _tiny(golden_dir, device)
    B, M, d = g["latent"].shape
    lat = torch.from_numpy(synth.uniform("golden/gpt/lat", (B, cfg.cond_latents, d), 0.5))
    emo = torch.from_numpy(synth.uniform("golden/gpt/emo", (B, d), 0.3))
    text = torch.from_numpy(synth.integers("golden/gpt/text2", (B, 7), 2, cfg.number_text_tokens))
    codes = torch.from_numpy(synth.integers("golden/gpt/codes2", (B, M), 0, cfg.start_mel_token))
    out = uv.forward(lat, text, torch.tensor([7, 7]), codes, torch.tensor([M, M]), emo_vec=emo)
    np.testing.assert_allclose(out.cpu().numpy(), g["latent"], rtol=0, atol=5e-5)


def test_decode_graph_is_kept_per_shape_and_replayed(device, golden_dir):
    """Greedy generations keep their instantiated decode-step graph per (workspace, batch, prompt length, max_new_tokens, penalty):
    the same call again re-captures nothing and returns the same codes; other shapes, an interleaved sampled call and a second
    stream each get their own; all equal the eager launches."""
    g, cfg, w, uv = _tiny(golden_dir, device)
    lib = _lib.load()
    B, L = g["greedy_text"].shape
    NEW = g["greedy_codes"].shape[1]
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5)).to(device)
    text = torch.from_numpy(g["greedy_text"])
    fake, emb, mask = uv.prepare_gpt_inputs(conds, text)
    kw = dict(stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb, repetition_penalty=10.0)
    run = lambda n=NEW, **k: uv.generate(fake, max_new_tokens=n, **{**kw, **k})[:, fake.shape[1]:].cpu().numpy()
    assert lib.idxtts_gpt_graph_cache_entries(uv._h) == 0
    a = run()
    assert lib.idxtts_gpt_graph_cache_entries(uv._h) == 1 and np.array_equal(a, g["greedy_codes"])
    assert np.array_equal(run(), a) and lib.idxtts_gpt_graph_cache_entries(uv._h) == 1          # replayed, not re-captured
    shorter = run(NEW - 3)                                                                        # another shape on the same workspace
    assert np.array_equal(shorter, run(NEW - 3, use_graph=False))
    n_after = lib.idxtts_gpt_graph_cache_entries(uv._h)
    assert n_after in (1, 2)                      # same workspace address -> the entry is replaced; a regrown workspace -> a second one
    sampled = run(do_sample=True, top_k=5, temperature=0.9, generator=torch.Generator().manual_seed(3))     # sampling: never cached
    assert sampled.shape[0] == B and lib.idxtts_gpt_graph_cache_entries(uv._h) == n_after
    assert np.array_equal(run(), a) and np.array_equal(run(repetition_penalty=2.0), run(repetition_penalty=2.0, use_graph=False))
    s2 = torch.cuda.Stream(device=device)
    with torch.cuda.stream(s2):                                                                   # its own workspace, its own graph
        b = run()
        s2.synchronize()
    assert np.array_equal(b, a)
    for _ in range(12):                                                                            # the cache is bounded
        run(NEW - 1 - (_ % 5))
    assert lib.idxtts_gpt_graph_cache_entries(uv._h) <= 8
    assert np.array_equal(run(), a)


def test_eos_and_padding_invariance(device):
    """Rows that stop keep emitting the stop token; a row decoded alone equals the same row left-padded in a batch
    (the reference's own property test, tests/padding_test.py:35-89); results equal the CPU oracle token for token."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/eos")
    w["mel_head.bias"] = w["mel_head.bias"].copy()
    w["mel_head.bias"][cfg.stop_mel_token] = 3.5
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L = 5, 9
    lat = torch.from_numpy(synth.uniform("t/gpt/eos/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/eos/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/eos/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 6:] = cfg.stop_text_token
    text[3, 2:] = cfg.stop_text_token
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=48, repetition_penalty=10.0)
    ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, 48, 10.0)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())
    c = codes.cpu().numpy()
    for row in c:
        hits = np.nonzero(row == cfg.stop_mel_token)[0]
        if len(hits):
            assert (row[hits[0]:] == cfg.stop_mel_token).all()
    solo, _ = uv.inference_speech(lat[3:4], text[3:4, :2], emo_vec=emo[3:4], max_generate_length=c.shape[1], repetition_penalty=10.0)
    n = min(solo.shape[1], c.shape[1])
    assert np.array_equal(solo.cpu().numpy()[0, :n], c[3, :n])


def test_latent_pass_split_bf16_vs_oracle(device):
    """A latent pass long enough (B*S >= 256 rows) for the split-bf16 GEMMs; greedy decode is unaffected (always fp32)."""
    from indextts_amd import _lib
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/lat16")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L, M = 3, 40, 100
    lat = torch.from_numpy(synth.uniform("t/gpt/lat16/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/lat16/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/lat16/text", (B, L), 2, cfg.number_text_tokens))
    codes = torch.from_numpy(synth.integers("t/gpt/lat16/codes", (B, M), 0, cfg.start_mel_token))
    ref = og.latent_forward(tw, cfg, lat, text, codes, emo)
    errs = {}
    try:
        for name, mode in (("f32", _lib.GEMM_F32), ("bf16x3", _lib.GEMM_BF16X3)):
            _lib.set_gemm_mode(mode)
            out = uv.forward(lat, text, torch.full((B,), L), codes, torch.full((B,), M), emo_vec=emo).cpu()
            errs[name] = (out - ref).abs().max().item()
    finally:
        _lib.set_gemm_mode(_lib.GEMM_BF16X3)
    assert errs["f32"] <= 1e-4 and errs["bf16x3"] <= 2e-3 and errs["bf16x3"] > 0


def test_greedy_batch_above_16_rows_vs_oracle(device):
    """B = 20 utterances: the decode GEMVs run with two row tiles (and the un-fused LayerNorm path)."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/b20")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L = 20, 8
    lat = torch.from_numpy(synth.uniform("t/gpt/b20/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/b20/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/b20/text", (B, L), 2, cfg.number_text_tokens))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=12, repetition_penalty=10.0)
    ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, 12, 10.0)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())


def test_greedy_48_rows_unsplit_attention_vs_oracle(device):
    """B x heads > 128 workgroups: the decode attention runs one workgroup per (utterance, head) (smaller batches split the
    keys of each head over several workgroups, which every other test here exercises); three row tiles in the GEMVs."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=256, heads=4, layers=2, number_text_tokens=300, number_mel_codes=258, start_mel_token=256,
                    stop_mel_token=257, max_mel_tokens=40, max_text_tokens=30, cond_latents=6)
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/b48")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L = 48, 7
    lat = torch.from_numpy(synth.uniform("t/gpt/b48/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/b48/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/b48/text", (B, L), 2, cfg.number_text_tokens))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=10, repetition_penalty=10.0)
    ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, 10, 10.0)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())


def test_greedy_wide_mlp_k_split_vs_oracle(device):
    """model_dim 256: mlp.c_proj has K = 1024, the shape class whose decode GEMV is split across workgroups
    (partial-sum slab + fixed-order combine, gemv_fx_ksb); codes must still equal the oracle's."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=256, heads=4, layers=2, number_text_tokens=300, number_mel_codes=258, start_mel_token=256,
                    stop_mel_token=257, max_mel_tokens=120, max_text_tokens=60, cond_latents=6)
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/wide")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L = 5, 9
    lat = torch.from_numpy(synth.uniform("t/gpt/wide/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/wide/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/wide/text", (B, L), 2, cfg.number_text_tokens))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=14, repetition_penalty=10.0)
    ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, 14, 10.0)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())


def test_sampled_codes_bit_exact_vs_hf_golden(device, golden_dir):
    """do_sample=True (HF warpers + torch.multinomial): fed the Exp(1) draws the golden run consumed, the HIP sampler must
    return HF's tokens exactly -- eager and through the replayed graph."""
    g, cfg, w, uv = _tiny(golden_dir, device)
    B, L = g["greedy_text"].shape
    NEW = g["sample_codes"].shape[1]
    temp, top_k, top_p = g["sample_params"]
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, cfg.model_dim), 0.5)).to(device)
    text = torch.from_numpy(g["greedy_text"])
    fake, emb, mask = uv.prepare_gpt_inputs(conds, text)
    noise = torch.from_numpy(g["sample_noise"])
    for graph in (False, True):
        out = uv.generate(fake, max_new_tokens=NEW, stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb,
                          repetition_penalty=10.0, use_graph=graph, do_sample=True, temperature=float(temp), top_k=int(top_k),
                          top_p=float(top_p), exp_noise=noise)
        assert np.array_equal(out[:, fake.shape[1]:].cpu().numpy(), g["sample_codes"]), f"graph={graph}"


@pytest.mark.parametrize("case", [("hf", 0.8, 30, 0.8), ("hf", 1.3, 5, 0.5), ("hf", 0.7, 50, 1.0), ("hf", 1.0, 0, 1.0), ("accel", 0.8, 0, 1.0)])
def test_sampling_modes_vs_oracle(device, case):
    """Both samplers (HF multinomial with warpers, the accel engine's Gumbel-max Sampler) on a ragged batch of 6 against the
    CPU oracle with the same Exp(1) draws; also through inference_speech with a seeded generator."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    sampler, temp, top_k, top_p = case
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/samp")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L, NEW = 6, 9, 14
    lat = torch.from_numpy(synth.uniform("t/gpt/samp/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/samp/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/samp/text", (B, L), 2, cfg.number_text_tokens))
    text[2, 6:] = cfg.stop_text_token
    text[5, 3:] = cfg.stop_text_token
    gen = torch.Generator().manual_seed(99)
    noise = torch.stack([torch.empty(B, cfg.number_mel_codes).exponential_(1, generator=gen) for _ in range(NEW)])
    ref = og.generate_sample(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, noise, 10.0, temp, top_k, top_p,
                             accel_sampler=(sampler == "accel"))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, do_sample=True,
                                   temperature=temp, top_k=top_k, top_p=top_p, sampler=sampler, exp_noise=noise)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())
    # the draws can also come from a seeded generator: same seed, same order of exponential_() calls, same tokens
    codes2, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, do_sample=True,
                                    temperature=temp, top_k=top_k, top_p=top_p, sampler=sampler,
                                    generator=torch.Generator().manual_seed(99))
    assert np.array_equal(codes2.cpu().numpy(), ref.numpy())


def test_full_size_greedy_and_latent_vs_oracle(device):
    """The real IndexTTS-2 GPT (1280 x 24 layers, 8194 codes): greedy codes of a ragged batch bit-exact against the fp32 CPU
    oracle, and the latent pass within tolerance -- the decode kernels at their production shapes (folded LayerNorm at
    d = 1280, K-split mlp.c_proj, 20 heads)."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig()
    w = weights.synth_gpt_weights(cfg, tag="bench/gpt")
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L, NEW = 2, 10, 8
    lat = torch.from_numpy(synth.uniform("t/gpt/full/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/full/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/full/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 6:] = cfg.stop_text_token
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, 10.0)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())
    n = ref.shape[1]
    lens = torch.tensor([n, n])
    got = uv.forward(lat, text, torch.tensor([L, 6]), codes.cpu(), lens, emo_vec=emo).cpu()
    # the batched latent pass masks the text padding so that every row equals its own B = 1 call (what infer_v2.py:816 runs)
    for b, tl in enumerate((L, 6)):
        with torch.no_grad():
            want = og.latent_forward(tw, cfg, lat[b:b + 1], text[b:b + 1, :tl], ref[b:b + 1], emo[b:b + 1])
        assert (got[b:b + 1] - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item()), b


# ---- compact weight storage (BASELINE configs[4]: fp8 GPT weights; `use_fp16`-like bf16) ------------------------------------
def _fp8_grid():
    """Every finite OCP e4m3fn value, restated independently of the library (bias 7, 3 mantissa bits, max 448)."""
    vals = []
    for c in range(127):
        e, m = (c >> 3) & 15, c & 7
        vals.append(m / 8 * 2.0 ** -6 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
    return np.array(vals, dtype=np.float64)


def _check_effective(orig, eff, cfg, fmt):
    """The read-back model is (ln = identity affine, Q(diag(g) W), b_ln . W + b) with Q = the format's rounding."""
    grid = _fp8_grid()
    for i in range(cfg.layers):
        p = f"gpt.h.{i}"
        for ln, proj in ((".ln_1", ".attn.c_attn"), (".ln_2", ".mlp.c_fc"), (None, ".attn.c_proj"), (None, ".mlp.c_proj")):
            W = orig[p + proj + ".weight"].astype(np.float64)
            if ln:
                g, b = orig[p + ln + ".weight"], orig[p + ln + ".bias"].astype(np.float64)
                assert np.all(eff[p + ln + ".weight"] == 1.0) and np.all(eff[p + ln + ".bias"] == 0.0)
                c = b @ W + orig[p + proj + ".bias"]
                assert np.abs(eff[p + proj + ".bias"] - c).max() <= 1e-6 * max(1.0, np.abs(c).max())
                W = (g[:, None] * orig[p + proj + ".weight"]).astype(np.float64)     # the fold is an fp32 product
            Q = eff[p + proj + ".weight"].astype(np.float64)
            if fmt == "bf16":
                assert np.all((eff[p + proj + ".weight"].view(np.uint32) & 0xffff) == 0)
                assert np.all(np.abs(Q - W) <= 2.0 ** -8 * np.abs(W))       # half an ulp of 8 significant bits
            else:
                mx = np.abs(W).max(axis=0)
                s = 2.0 ** np.ceil(np.log2(np.maximum(mx, 1e-30) / 448.0))
                q = np.abs(Q) / s
                assert np.all(np.isin(q, grid)), "not an e4m3 value x power-of-two column scale"
                # nearest grid point (ties may go either way here; the library rounds them to the even code)
                assert np.all(np.abs(q - np.abs(W) / s) <= np.abs(grid[None, None, :] - (np.abs(W) / s)[..., None]).min(axis=-1) + 1e-12)


@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
def test_compact_weights_generate_and_latent_vs_oracle_on_the_same_model(device, fmt):
    """Weights stored as bf16 / fp8-e4m3 (+ power-of-two column scale): the rounded model is read back through the C ABI and
    handed to the fp32 CPU oracle -- greedy codes bit-exact, latent within the fp32 tolerance; and the read-back model is
    checked to be exactly the documented rounding of the original one.  Compact weights bring the bf16 KV cache with them
    (the reference's `use_fp16` halves both): the oracle rounds keys / values the same way (kv_round)."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=128, heads=2, layers=3, number_mel_codes=210, number_text_tokens=60, start_mel_token=208, stop_mel_token=209,
                    max_mel_tokens=60, max_text_tokens=30)
    w = weights.synth_gpt_weights(cfg, tag=f"t/gpt/q/{fmt}")
    uv = UnifiedVoice(w, cfg, device=device, weight_format=fmt, keep_effective=True)
    assert uv.kv_format == "bf16"
    eff = uv.effective_state_dict
    _check_effective(w, eff, cfg, fmt)
    tw = {k: torch.from_numpy(v) for k, v in eff.items()}
    B, L, NEW = 3, 9, 40
    lat = torch.from_numpy(synth.uniform("t/gpt/q/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/q/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/q/text", (B, L), 2, cfg.number_text_tokens))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, 10.0, kv_round=True)
    assert np.array_equal(codes.cpu().numpy(), ref.numpy())
    uv.set_kv_format("f32")                                 # compact weights with the fp32 cache: the oracle without the rounding
    codes32, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref32 = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, 10.0)
    assert np.array_equal(codes32.cpu().numpy(), ref32.numpy())
    uv.set_kv_format("bf16")
    n = ref.shape[1]
    got = uv.forward(lat, text, torch.full((B,), L), codes.cpu(), torch.full((B,), n), emo_vec=emo).cpu()
    with torch.no_grad():
        want = og.latent_forward(tw, cfg, lat, text, ref, emo)
    assert (got - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())


def test_fp8_full_size_single_utterance_vs_oracle(device):
    """configs[4] shape: the real GPT, ONE utterance, emotion vector mixed in, fp8 weight streams, graph-replayed decode --
    codes bit-exact against the oracle run on the read-back fp8 model."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig()
    w = weights.synth_gpt_weights(cfg, tag="bench/gpt")
    uv = UnifiedVoice(w, cfg, device=device, weight_format="fp8", keep_effective=True)
    tw = {k: torch.from_numpy(v) for k, v in uv.effective_state_dict.items()}
    L, NEW = 12, 10
    lat = torch.from_numpy(synth.uniform("t/gpt/q8/lat", (1, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/q8/emo", (1, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/q8/text", (1, L), 2, cfg.number_text_tokens))
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, 10.0, kv_round=uv.kv_format == "bf16")
    assert uv.kv_format == "bf16" and np.array_equal(codes.cpu().numpy(), ref.numpy())


def test_maximum_text_length_and_long_decode_vs_oracle(device):
    """Edge of the position tables: a 600-token text (max_text_tokens) and several hundred generated codes on a narrow model, so
    that the decode attention walks > 1024 cached keys (more than two passes of its score / P.V loops, and, with B = 2, key pieces
    of > 64 keys per workgroup) -- greedy codes bit-exact against the oracle."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=128, heads=2, layers=2, number_mel_codes=130, number_text_tokens=90, start_mel_token=128, stop_mel_token=129,
                    max_mel_tokens=1815, max_text_tokens=600, cond_latents=8)
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/maxlen")
    w["mel_head.bias"][cfg.stop_mel_token] = -1e4          # never stop: the full 450 steps
    uv = UnifiedVoice(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L, NEW = 2, 600, 450
    lat = torch.from_numpy(synth.uniform("t/gpt/maxlen/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/maxlen/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/maxlen/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 37:] = cfg.stop_text_token                      # a short row next to the maximal one (left-padded prompt)
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref = og.generate_greedy(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, 10.0)
    assert codes.shape[1] == NEW and np.array_equal(codes.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("B,heads", [(2, 2), (9, 16)])
def test_bf16_kv_cache_long_decode_vs_oracle(device, B, heads):
    """idxtts_gpt_set_kv_format(1) on fp32 weights: keys / values rounded to bf16 when produced (prefill and decode), fp32 arithmetic --
    greedy codes bit-exact against the oracle that rounds the same way, over several hundred cached keys (more than one pass of the
    score loop, every lane of the 8-lane value groups), with the key range split over workgroups (B * heads = 4) and not (144);
    a ragged batch (left padding), switching the format back and forth on one context (cached decode graphs are keyed by it)."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=64 * heads, heads=heads, layers=2, number_mel_codes=130, number_text_tokens=90, start_mel_token=128, stop_mel_token=129,
                    max_mel_tokens=700, max_text_tokens=300, cond_latents=8)
    w = weights.synth_gpt_weights(cfg, tag=f"t/gpt/kv16/{heads}")
    w["mel_head.bias"][cfg.stop_mel_token] = -1e4          # never stop
    uv = UnifiedVoice(w, cfg, device=device, kv_format="bf16")
    assert uv.kv_format == "bf16" and uv.weight_format == "f32"
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    L, NEW = 290, 300 if B == 2 else 40
    lat = torch.from_numpy(synth.uniform("t/gpt/kv16/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/kv16/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/kv16/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 23:] = cfg.stop_text_token
    conds = og.conds_latent(tw, cfg, lat, emo)
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    with torch.no_grad():
        ref16 = og.generate_greedy(tw, cfg, conds, text, NEW, 10.0, kv_round=True)
        ref32 = og.generate_greedy(tw, cfg, conds, text, NEW, 10.0)
    assert codes.shape[1] == NEW and np.array_equal(codes.cpu().numpy(), ref16.numpy())
    uv.set_kv_format("f32")
    codes32, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    assert np.array_equal(codes32.cpu().numpy(), ref32.numpy())
    uv.set_kv_format("bf16")
    again, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
    assert torch.equal(again, codes)


def test_bf16_kv_cache_beam_sample_vs_oracle(device):
    """Beam search and beam-sample (the reference's default mode) on the bf16 cache: the KV rows of the generated positions are
    re-indexed by beam ancestry in 16-byte granules of the bf16 layout -- tokens bit-exact against oracle.gpt.generate_beam with kv_round."""
    from indextts_amd.gpt import UnifiedVoice
    from oracle import gpt as og
    cfg = GPTConfig(model_dim=128, heads=2, layers=2, number_mel_codes=70, number_text_tokens=40, start_mel_token=68, stop_mel_token=69,
                    max_mel_tokens=60, max_text_tokens=30, cond_latents=4)
    w = weights.synth_gpt_weights(cfg, tag="t/gpt/kv16/beam")
    w["mel_head.bias"][cfg.stop_mel_token] += 3.0
    uv = UnifiedVoice(w, cfg, device=device, kv_format="bf16")
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    B, L, NEW, NB = 2, 7, 24, 3
    lat = torch.from_numpy(synth.uniform("t/gpt/kv16/beam/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/kv16/beam/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/kv16/beam/text", (B, L), 2, cfg.number_text_tokens))
    g = torch.Generator().manual_seed(11)
    noise = torch.empty(NEW, B, NB * cfg.number_mel_codes).exponential_(1.0, generator=g)
    for do_sample in (True, False):
        codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, num_beams=NB, do_sample=do_sample, top_p=0.8, top_k=30,
                                       temperature=0.8, repetition_penalty=10.0, length_penalty=0.0, exp_noise=noise)
        with torch.no_grad():
            want = og.generate_beam(tw, cfg, og.conds_latent(tw, cfg, lat, emo), text, NEW, noise, num_beams=NB, do_sample=do_sample, kv_round=True)
        got = codes.cpu().numpy()
        assert got.shape == tuple(want.shape) and np.array_equal(got, want.numpy()), do_sample


@pytest.fixture
def plane_from_5_rows():
    """The plane GEMV from 5 decode rows on (default: 17), so that its one-row-tile geometry is exercised too."""
    from indextts_amd import _lib
    _lib.set_decode_plane_rows(5)
    yield
    _lib.set_decode_plane_rows(0)
    assert _lib.get_decode_plane_rows() == 17


def _pl_model(device, fmt, heads=2, kv="bf16", tag="t/gpt/pl", V=210, stop_bias=0.0):
    from indextts_amd.gpt import UnifiedVoice
    cfg = GPTConfig(model_dim=64 * heads, heads=heads, layers=3, number_mel_codes=V, number_text_tokens=60, start_mel_token=V - 2, stop_mel_token=V - 1,
                    max_mel_tokens=80, max_text_tokens=30)
    w = weights.synth_gpt_weights(cfg, tag=f"{tag}/{fmt}/{heads}")
    w["mel_head.bias"] = w["mel_head.bias"].copy()
    w["mel_head.bias"][cfg.stop_mel_token] += stop_bias
    uv = UnifiedVoice(w, cfg, device=device, weight_format=fmt, keep_effective=True, kv_format=kv)
    return cfg, uv, {k: torch.from_numpy(v) for k, v in uv.effective_state_dict.items()}


def _assert_equal_or_near_tie(got, ref, ref_logits, fake, tol, penalty=10.0):
    """Greedy codes equal the oracle's, or a row leaves the oracle's sequence at a step where the ORACLE's own margin between its token
    and the row's token is below `tol` (the bf16 KV cache adds ~2e-3 of noise to a logit: a key / value one ulp apart between two fp32
    summation orders may round to the other bf16 neighbour; oracle/parity.py measures it at full size)."""
    from oracle import gpt as og
    got, ref = np.asarray(got), np.asarray(ref)
    n = min(got.shape[1], ref.shape[1])
    flips = 0
    for b in range(ref.shape[0]):
        d = np.nonzero(got[b, :n] != ref[b, :n])[0]
        if len(d) == 0:
            continue
        s = int(d[0])
        ids = torch.cat([fake[b], torch.from_numpy(ref[b, :s].astype(np.int64))])[None]
        sc = og.repetition_penalty(ids, ref_logits[b, s][None].float(), penalty)[0]
        margin = float(sc[int(ref[b, s])] - sc[int(got[b, s])])
        assert 0.0 <= margin <= tol, f"row {b} leaves the oracle at step {s} where its margin is {margin:.3e} (> {tol})"
        flips += 1
    return flips


@pytest.mark.parametrize("fmt,B,heads", [("bf16", 5, 2), ("bf16", 16, 2), ("bf16", 20, 3), ("bf16", 48, 2), ("bf16", 64, 2), ("fp8", 6, 2), ("fp8", 33, 3), ("fp8", 64, 2)])
def test_plane_gemv_decode_rows_vs_oracle(device, plane_from_5_rows, fmt, B, heads):
    """Compact weight streams with more than 4 decode rows run the decode step on the bf16-MFMA plane GEMV (csrc/gemv_pl.hip: activations
    as three bf16 planes, K split over waves and workgroups, LayerNorm statistics handed over per 16 columns, sample + embed + advance
    in one launch): 1-4 row tiles, both formats, d = 128 / 192 (K not a multiple of the workgroup's K part), ragged texts, a stop
    bias so that rows finish at different steps -- greedy codes bit-exact against the oracle on the read-back rounded model, per-step
    logits within the fp32 tolerance, graph replay == eager launches, and both KV formats."""
    from oracle import gpt as og
    cfg, uv, tw = _pl_model(device, fmt, heads, stop_bias=1.2)
    L, NEW = 9, 30
    lat = torch.from_numpy(synth.uniform("t/gpt/pl/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/pl/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/pl/text", (B, L), 2, cfg.number_text_tokens))
    for b in range(B):
        text[b, L - (b % 4):] = cfg.stop_text_token
    conds = og.conds_latent(tw, cfg, lat, emo)
    fake = og.prepare_gpt_inputs(tw, cfg, conds, text)[0]
    for kv in ("bf16", "f32"):
        uv.set_kv_format(kv)
        with torch.no_grad():
            ref, ref_logits = og.generate_greedy(tw, cfg, conds, text, NEW, 10.0, return_logits=True, kv_round=kv == "bf16")
        codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0)
        if kv == "f32":      # nothing is rounded between the weights and the logits: bit-exact codes
            assert np.array_equal(codes.cpu().numpy(), ref.numpy()), kv
        else:                # bf16 cache: equal, or parted at a near-tie of the oracle (at most a row or two of a batch)
            assert _assert_equal_or_near_tie(codes.cpu().numpy(), ref.numpy(), ref_logits, fake, 1e-2) <= max(1, B // 16)
        eager, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, use_graph=False)
        assert torch.equal(eager, codes)
        # per-step logits on the ORACLE's tokens (teacher-forced: a row that parted at a tie would compare two different sequences)
        n = ref.shape[1]
        forced = torch.full((B, NEW), cfg.stop_mel_token, dtype=torch.long)
        forced[:, :n] = ref
        _, _, logits = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, return_logits=True, forced_codes=forced)
        tol = 5e-5 if kv == "f32" else 1e-2
        assert (logits.cpu()[:, :n] - ref_logits).abs().max().item() <= tol, kv
    uv.set_kv_format("bf16")


def test_plane_gemv_rows_do_not_depend_on_the_batch(device, plane_from_5_rows):
    """Row b of a 60-row decode (4 row tiles: 8 column tiles per workgroup) equals, bit for bit in every logit, the same utterance decoded
    in a 44-row and a 46-row batch (3 row tiles, 4 column tiles per workgroup; different positions inside the tiles): the plane GEMV's
    arithmetic per row depends neither on the batch nor on the geometry the batch selects --
    what lets serving.BatchPipeline merge waiting requests into one decode.  (3 heads: every batch here has more than 128 (utterance,
    head) pairs, so the decode attention never splits its keys -- the one kernel whose summation order follows the batch size.)"""
    cfg, uv, tw = _pl_model(device, "bf16", 3)
    B, L, NEW = 60, 8, 16
    lat = torch.from_numpy(synth.uniform("t/gpt/plb/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/plb/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/plb/text", (B, L), 2, cfg.number_text_tokens))
    full, _, lg = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, return_logits=True)
    for lo, hi in ((0, 44), (14, 60)):
        part, _, lp = uv.inference_speech(lat[lo:hi], text[lo:hi], emo_vec=emo[lo:hi], max_generate_length=NEW, repetition_penalty=10.0, return_logits=True)
        n = min(part.shape[1], full.shape[1])
        assert torch.equal(part[:, :n], full[lo:hi, :n]) and torch.equal(lp[:, :n], lg[lo:hi, :n]), (lo, hi)


@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
def test_plane_gemv_sampling_and_beams_vs_oracle(device, plane_from_5_rows, fmt):
    """The plane-GEMV decode step under the other decoding modes (separate embed / sampler / advance launches): HF multinomial sampling
    on 6 rows and beam-sample / beam search on 4 utterances x 3 beams = 12 rows, tokens bit-exact against the oracle."""
    from oracle import gpt as og
    cfg, uv, tw = _pl_model(device, fmt, 2, tag="t/gpt/pls", V=90)
    B, L, NEW, NB = 6, 9, 14, 3
    lat = torch.from_numpy(synth.uniform("t/gpt/pls/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
    emo = torch.from_numpy(synth.uniform("t/gpt/pls/emo", (B, cfg.model_dim), 0.3))
    text = torch.from_numpy(synth.integers("t/gpt/pls/text", (B, L), 2, cfg.number_text_tokens))
    gen = torch.Generator().manual_seed(3)
    noise = torch.stack([torch.empty(B, cfg.number_mel_codes).exponential_(1, generator=gen) for _ in range(NEW)])
    conds = og.conds_latent(tw, cfg, lat, emo)
    with torch.no_grad():
        want = og.generate_sample(tw, cfg, conds, text, NEW, noise, 10.0, 0.8, 30, 0.8, kv_round=True)
    codes, _ = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, do_sample=True, top_p=0.8, top_k=30, temperature=0.8,
                                   repetition_penalty=10.0, exp_noise=noise)
    assert np.array_equal(codes.cpu().numpy(), want.numpy())
    Bb = 4
    bnoise = torch.empty(NEW, Bb, NB * cfg.number_mel_codes).exponential_(1.0, generator=gen)
    for do_sample in (True, False):
        with torch.no_grad():
            wb = og.generate_beam(tw, cfg, conds[:Bb], text[:Bb], NEW, bnoise, num_beams=NB, do_sample=do_sample, kv_round=True)
        cb, _ = uv.inference_speech(lat[:Bb], text[:Bb], emo_vec=emo[:Bb], max_generate_length=NEW, num_beams=NB, do_sample=do_sample, top_p=0.8, top_k=30,
                                    temperature=0.8, repetition_penalty=10.0, length_penalty=0.0, exp_noise=bnoise)
        assert cb.shape == tuple(wb.shape) and np.array_equal(cb.cpu().numpy(), wb.numpy()), do_sample
