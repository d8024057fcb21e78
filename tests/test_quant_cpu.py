"""CPU: host helpers of the fp8 weight format (OCP e4m3fn) exported by the C ABI, against an independent restatement."""
import numpy as np

from indextts_amd import _lib


def _grid():
    vals = []
    for c in range(127):
        e, m = (c >> 3) & 15, c & 7
        vals.append(m / 8 * 2.0 ** -6 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
    return np.array(vals, dtype=np.float64)


def test_fp8_decode_all_codes():
    lib = _lib.load()
    g = _grid()
    for c in range(256):
        v = lib.idxtts_fp8_e4m3_decode(c)
        if (c & 0x7f) == 0x7f:
            assert np.isnan(v)
        else:
            assert v == (-1.0 if c & 0x80 else 1.0) * g[c & 0x7f], c
    assert g.max() == 448.0 and g[1] == 2.0 ** -9


def test_fp8_encode_is_nearest_even_and_saturates():
    lib = _lib.load()
    g = _grid()
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.uniform(-460, 460, 3000), rng.uniform(-0.05, 0.05, 2000), rng.standard_normal(2000) * 8,
                           g, -g, (g[:-1] + g[1:]) / 2, [0.0, -0.0, 1e-30, 500.0, -1e9]]).astype(np.float32)
    for v in vals:
        c = lib.idxtts_fp8_e4m3_encode(float(v))
        assert (c & 0x7f) != 0x7f                                    # never NaN
        a = abs(float(v))
        d = np.abs(g - min(a, 448.0))
        best = np.flatnonzero(d == d.min())
        want = best[0] if len(best) == 1 else [b for b in best if b % 2 == 0][0]      # tie -> even code
        assert (c & 0x7f) == want, (v, c, want)
        if a > 0:
            assert bool(c & 0x80) == bool(v < 0)


def test_bf16_weight_rounding_transform_on_host():
    """idxtts_gpt_quantize_weights(format 1) is a host transform of the staged tensors: LayerNorm folded into c_attn / c_fc, every
    linear weight rounded to bf16, read back under the reference's keys -- no GPU involved before finalize."""
    import ctypes
    from ctypes import c_void_p
    from indextts_amd import weights
    from indextts_amd.config import GPTConfig
    lib = _lib.load()
    cfg = GPTConfig(model_dim=64, heads=1, layers=2, number_mel_codes=40, number_text_tokens=30, start_mel_token=38, stop_mel_token=39,
                    max_mel_tokens=20, max_text_tokens=10, cond_latents=4)
    w = weights.synth_gpt_weights(cfg, tag="t/quant/cpu")
    c = _lib.GPTConfigC(cfg.model_dim, cfg.heads, cfg.layers, cfg.number_mel_codes, cfg.number_text_tokens, cfg.start_mel_token,
                        cfg.stop_mel_token, cfg.mel_pos_len, cfg.text_pos_len)
    h = c_void_p()
    _lib.check(lib.idxtts_gpt_create(ctypes.byref(c), ctypes.byref(h)))
    try:
        for name, arr in w.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(lib.idxtts_ctx_load_tensor(h, name.encode(), c_void_p(a.ctypes.data), shape, a.ndim))
        _lib.check(lib.idxtts_gpt_quantize_weights(h, 1))
        for i in range(cfg.layers):
            p = f"gpt.h.{i}"
            for ln, proj in ((".ln_1", ".attn.c_attn"), (".ln_2", ".mlp.c_fc")):
                g, b = w[p + ln + ".weight"], w[p + ln + ".bias"].astype(np.float64)
                W = w[p + proj + ".weight"]
                q = _lib.get_tensor(h, p + proj + ".weight", W.shape)
                folded = (g[:, None] * W).astype(np.float32)
                assert np.all((q.view(np.uint32) & 0xffff) == 0)
                assert np.all(np.abs(q.astype(np.float64) - folded) <= 2.0 ** -8 * np.abs(folded))
                assert np.all(_lib.get_tensor(h, p + ln + ".weight", g.shape) == 1.0)
                assert np.all(_lib.get_tensor(h, p + ln + ".bias", g.shape) == 0.0)
                cb = _lib.get_tensor(h, p + proj + ".bias", (W.shape[1],))
                want = b @ W.astype(np.float64) + w[p + proj + ".bias"]
                assert np.abs(cb - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
            for proj in (".attn.c_proj", ".mlp.c_proj"):
                W = w[p + proj + ".weight"]
                q = _lib.get_tensor(h, p + proj + ".weight", W.shape)
                assert np.all((q.view(np.uint32) & 0xffff) == 0) and np.all(np.abs(q - W) <= 2.0 ** -8 * np.abs(W))
        emb = _lib.get_tensor(h, "mel_embedding.weight", w["mel_embedding.weight"].shape)
        assert np.array_equal(emb, w["mel_embedding.weight"])            # embeddings are gathers, not streams: untouched
        assert lib.idxtts_gpt_quantize_weights(h, 1) != 0                  # once only
    finally:
        lib.idxtts_ctx_destroy(h)
