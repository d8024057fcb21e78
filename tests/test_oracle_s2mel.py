"""CPU: the s2mel oracle (oracle/s2mel.py) against golden vectors produced by the reference's own
MyModel / FactorizedVectorQuantize (tests/golden/make_golden.py::make_s2mel)."""
import os

import numpy as np
import torch

from indextts_amd import synth, weights
from indextts_amd.config import S2MelConfig
from oracle import s2mel as osm

TOL = 2e-5


def _setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "s2mel.npz"))
    cfg = S2MelConfig.tiny()
    w = {k: torch.from_numpy(v) for k, v in weights.synth_s2mel_weights(cfg, tag="golden/s2mel").items()}
    return g, cfg, w


def test_gpt_layer_and_vq2emb(golden_dir):
    g, cfg, w = _setup(golden_dir)
    lat = torch.from_numpy(synth.uniform("golden/s2mel/latent", (2, 9, cfg.gpt_dim), 1.0))
    np.testing.assert_allclose(osm.gpt_layer(w, lat).numpy(), g["gpt_layer"], rtol=0, atol=TOL)
    codes = torch.from_numpy(synth.integers("golden/s2mel/codes", (2, 9), 0, cfg.codebook_size))
    np.testing.assert_allclose(osm.vq2emb(w, codes).numpy(), g["vq2emb"], rtol=0, atol=TOL)


def test_length_regulator(golden_dir):
    g, cfg, w = _setup(golden_dir)
    for tag, M in (("a", 9), ("b", 20)):
        S = torch.from_numpy(synth.uniform(f"golden/s2mel/S_{tag}", (1, M, cfg.lr_in_channels), 1.0))
        ylens = (torch.LongTensor([M]) * 1.72).long()
        out = osm.length_regulator(w, cfg, S, ylens)
        assert out.shape == g[f"lr_{tag}"].shape
        np.testing.assert_allclose(out.numpy(), g[f"lr_{tag}"], rtol=0, atol=TOL)


def test_dit_forward_with_key_padding(golden_dir):
    g, cfg, w = _setup(golden_dir)
    T = 37
    x = torch.from_numpy(synth.uniform("golden/s2mel/dit/x", (2, cfg.in_channels, T), 1.0))
    px = torch.from_numpy(synth.uniform("golden/s2mel/dit/prompt", (2, cfg.in_channels, T), 1.0))
    px[..., 12:] = 0
    st = torch.from_numpy(synth.uniform("golden/s2mel/dit/style", (2, cfg.style_dim), 1.0))
    mu = torch.from_numpy(synth.uniform("golden/s2mel/dit/mu", (2, T, cfg.content_dim), 1.0))
    out = osm.dit_forward(w, cfg, x, px, torch.LongTensor([T, T - 6]), torch.tensor([0.35, 0.35]), st, mu)
    np.testing.assert_allclose(out.numpy(), g["dit"], rtol=0, atol=5e-5)


def test_cfm_euler_with_cfg(golden_dir):
    g, cfg, w = _setup(golden_dir)
    Tp, Tg = 11, 23
    T = Tp + Tg
    z = torch.from_numpy(synth.uniform("golden/s2mel/cfm/z", (1, cfg.in_channels, T), 1.7))
    mu = torch.from_numpy(synth.uniform("golden/s2mel/cfm/mu", (1, T, cfg.content_dim), 1.0))
    prompt = torch.from_numpy(synth.uniform("golden/s2mel/cfm/prompt", (1, cfg.in_channels, Tp), 1.0))
    st = torch.from_numpy(synth.uniform("golden/s2mel/cfm/style", (1, cfg.style_dim), 1.0))
    out = osm.cfm_inference(w, cfg, mu, torch.LongTensor([T]), prompt, st, z, 3, 0.7)
    np.testing.assert_allclose(out.numpy(), g["cfm"], rtol=0, atol=1e-4)
    assert (out[..., :Tp] == 0).all()          # prompt region is re-zeroed every step (flow_matching.py:113)
    # batched generalisation: two identical rows give the single-row answer (rows are independent)
    out2 = osm.cfm_inference(w, cfg, mu.repeat(2, 1, 1), torch.LongTensor([T, T]), prompt.repeat(2, 1, 1), st.repeat(2, 1),
                             z.repeat(2, 1, 1), 3, 0.7)
    np.testing.assert_allclose(out2[1].numpy(), g["cfm"][0], rtol=0, atol=1e-4)
