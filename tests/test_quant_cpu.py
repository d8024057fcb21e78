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
