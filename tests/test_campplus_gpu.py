"""GPU: the HIP CAMPPlus speaker encoder (`idxtts_campplus_forward`) against the reference-generated fixtures (real configuration) and
against the CPU oracle on a 15 s prompt's worth of frames."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import CamPPlusConfig

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "campplus.npz"))


@pytest.fixture(scope="module")
def model(device):
    from indextts_amd.campplus import CAMPPlus
    cfg = CamPPlusConfig()
    w = weights.synth_campplus_weights(cfg, tag="golden/campplus")
    return cfg, w, CAMPPlus(w, cfg, device=device)


def test_style_matches_reference_fixture(device, model):
    cfg, w, cp = model
    for tag in ("a", "b"):
        got = cp(torch.from_numpy(G[f"feat_{tag}"])).cpu().numpy()
        want = G[f"style_{tag}"]
        scale = max(1.0, np.abs(want).max())
        assert got.shape == want.shape == (1, 192)
        assert np.abs(got - want).max() <= 5e-4 * scale, (tag, np.abs(got - want).max(), scale)
    both = cp(torch.from_numpy(np.concatenate([G["feat_b"], G["feat_b"][:, ::-1].copy()]))).cpu().numpy()      # a batch = its rows one by one
    assert np.abs(both[0] - G["style_b"][0]).max() <= 5e-4 * max(1.0, np.abs(G["style_b"]).max())
    with pytest.raises(ValueError):
        cp(torch.zeros(1, 4, 80))


def test_fifteen_second_prompt_vs_oracle(device, model):
    from oracle import campplus as ocp
    cfg, w, cp = model
    T = 1498                                     # frames of a 15 s prompt at 10 ms
    feat = torch.from_numpy(synth.uniform("t/campplus/feat", (1, T, 80), 2.0))
    feat = feat - feat.mean(dim=1, keepdim=True)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        want = ocp.forward(tw, cfg, feat).numpy()
    got = cp(feat).cpu().numpy()
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= 5e-4 * scale, (np.abs(got - want).max(), scale)
