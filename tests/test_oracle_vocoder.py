"""CPU: the vocoder oracle (oracle/vocoder.py) against the golden vectors produced by the
reference's own modules (tests/golden/make_golden.py -> vocoder.npz)."""
import os

import numpy as np
import torch

from indextts_amd import synth, weights
from indextts_amd.config import BigVGANConfig
from oracle import vocoder as ov


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "vocoder.npz"))


def test_filter_matches_reference_buffers(golden_dir):
    g = _golden(golden_dir)
    f = ov.aa_filter().numpy()
    assert np.array_equal(f, g["up_filter"])
    assert np.array_equal(f, g["down_filter"])
    assert abs(f.sum() - 1.0) < 1e-6


def test_activation1d_matches_reference(golden_dir):
    g = _golden(golden_dir)
    for tag in "abcd":   # includes T=1, 7, 13: replicate pads dominate
        B, C, T = (int(v) for v in g[f"act_{tag}_shape"])
        la = synth.uniform(f"golden/act/{tag}/alpha", (C,), 0.8)
        lb = synth.uniform(f"golden/act/{tag}/beta", (C,), 0.8, offset=0.2)
        x = synth.uniform(f"golden/act/{tag}/x", (B, C, T), 3.0)
        y = ov.activation1d(torch.from_numpy(x), torch.from_numpy(la), torch.from_numpy(lb)).numpy()
        np.testing.assert_allclose(y, g[f"act_{tag}_y"], rtol=0, atol=3e-6)


def test_bigvgan_matches_reference(golden_dir):
    g = _golden(golden_dir)
    for tag in ("w64", "w128"):
        c0, B, Tm = (int(v) for v in g[f"bigvgan_{tag}_cfg"])
        cfg = BigVGANConfig.tiny(c0)
        w = weights.synth_bigvgan_weights(cfg, tag=f"golden/bigvgan/{tag}")
        mel = torch.from_numpy(weights.synth_mel(f"golden/bigvgan/{tag}/mel", B, cfg.num_mels, Tm))
        pre = ov.bigvgan_forward(w, cfg, mel, clamp=False).numpy()
        np.testing.assert_allclose(pre, g[f"bigvgan_{tag}_preclamp"], rtol=0, atol=5e-6)
        wav = ov.bigvgan_forward(w, cfg, mel, clamp=True).numpy()
        np.testing.assert_allclose(wav, g[f"bigvgan_{tag}_wav"], rtol=0, atol=5e-6)
        assert np.abs(g[f"bigvgan_{tag}_preclamp"]).std() > 0.05   # the fixture is not degenerate


def test_synth_is_deterministic():
    a = synth.uniform("x/y", (3, 5), 2.0)
    b = synth.uniform("x/y", (3, 5), 2.0)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(float(a.reshape(-1)[0]) - float(synth.uniform("x/y", (1,), 2.0)[0])) == 0.0
    assert np.abs(a).max() <= 2.0
    i = synth.integers("ids", (4, 7), 2, 12000)
    assert i.min() >= 2 and i.max() < 12000 and i.dtype == np.int64
