"""CPU: the oracle's CAMPPlus (oracle/campplus.py) against fixtures the reference's own class produced on the synthetic weights
(tests/golden/make_golden.py::make_campplus)."""
import os

import numpy as np
import torch

from indextts_amd import weights
from indextts_amd.config import CamPPlusConfig
from oracle import campplus as ocp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "campplus.npz"))


def test_style_vectors_match_reference():
    cfg = CamPPlusConfig()
    w = {k: torch.from_numpy(v) for k, v in weights.synth_campplus_weights(cfg, tag="golden/campplus").items()}
    with torch.no_grad():
        assert np.abs(ocp.fcm(torch.from_numpy(G["feat_b"]).permute(0, 2, 1), w).numpy() - G["fcm_b"]).max() <= 1e-4
        for tag in ("a", "b"):
            got = ocp.forward(w, cfg, torch.from_numpy(G[f"feat_{tag}"])).numpy()
            scale = max(1.0, np.abs(G[f"style_{tag}"]).max())
            assert np.abs(got - G[f"style_{tag}"]).max() <= 1e-4 * scale, tag
