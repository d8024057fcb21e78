"""CPU: the host-side prompt features (indextts_amd/features.py) against the container's own SeamlessM4TFeatureExtractor -- the
class the reference instantiates (infer_v2.py:201) -- on synthetic audio."""
import numpy as np
import pytest

from indextts_amd import features, synth


def _wave(tag, n):
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1370 * t + 0.5) + 0.05 * synth.uniform(tag, (n,), 1.0)
    return (x * np.hanning(n) ** 0.1).astype(np.float32)


def test_seamless_m4t_features_match_transformers():
    from transformers import SeamlessM4TFeatureExtractor
    fe = SeamlessM4TFeatureExtractor()              # the defaults are facebook/w2v-bert-2.0's preprocessor_config
    waves = [_wave("t/feat/a", 16000 * 2 + 123), _wave("t/feat/b", 16000 + 7), _wave("t/feat/c", 9999)]
    want = fe(waves, sampling_rate=16000, return_tensors="np")
    got = features.seamless_m4t_features(waves)
    assert got["input_features"].shape == want["input_features"].shape
    assert np.array_equal(got["attention_mask"], want["attention_mask"])
    m = want["attention_mask"].astype(bool)
    err = np.abs(got["input_features"] - want["input_features"])
    assert err[m].max() <= 2e-4, err[m].max()
    assert err.max() <= 2e-4
    one = features.seamless_m4t_features(waves[0])
    w1 = fe(waves[0], sampling_rate=16000, return_tensors="np")
    assert np.abs(one["input_features"] - w1["input_features"]).max() <= 2e-4 and np.array_equal(one["attention_mask"], w1["attention_mask"])


def test_kaldi_fbank_properties():
    x = _wave("t/feat/d", 16000)
    f = features.kaldi_fbank(x)
    assert f.shape == (1 + (16000 - 400) // 160, 80) and np.isfinite(f).all()
    # scaling the samples shifts every log energy by 2 log(scale) (away from the floor): what makes the CAMPPlus input, which
    # subtracts the mean over time, independent of the 16-bit scaling the w2v-bert extractor applies
    g = features.kaldi_fbank(x, scale=float(2 ** 15))
    assert np.abs((g - f) - 2 * np.log(2.0 ** 15)).max() <= 1e-3
    assert features.kaldi_fbank(x[:300]).shape == (0, 80)
    with pytest.raises(ValueError):
        features.seamless_m4t_features(x[:500])
