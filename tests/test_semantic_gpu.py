"""GPU: the HIP semantic-feature model (w2v-bert layers behind `idxtts_w2vbert_forward`) against the transformers-generated
fixture (tiny sizes, ragged batch) and against the CPU oracle at the real widths (1024 / 4096 / 16 heads / k31, 2 layers)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import W2VBertConfig

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "w2vbert.npz"))


def test_get_emb_matches_transformers_fixture(device):
    from indextts_amd.semantic import SemanticModel
    cfg = W2VBertConfig.tiny()
    w = weights.synth_w2vbert_weights(cfg, tag="golden/w2vbert")
    sm = SemanticModel(w, cfg, device=device)
    feats, mask, lens = torch.from_numpy(G["feats"]), torch.from_numpy(G["mask"]), G["lens"]
    got = sm(feats, mask).cpu().numpy()
    for b in range(feats.shape[0]):
        n = int(lens[b])
        assert np.abs(got[b, :n] - G["emb_ragged"][b, :n]).max() <= 2e-4, b
    solo = sm(feats[1:2, :22]).cpu().numpy()
    assert np.abs(solo - G["emb_row1_alone"]).max() <= 2e-4
    # a ragged row equals its own unpadded run (what masking padded frames as keys and conv inputs means)
    assert np.abs(got[1, :22] - solo[0]).max() <= 1e-5
    with pytest.raises(ValueError):
        sm(feats, torch.flip(mask, dims=[1]))          # left padding is not what the feature extractor produces


def test_full_width_layers_vs_oracle(device):
    """hidden 1024, 16 heads, ffn 4096, distance embedding 64 left / 8 right, causal depthwise k31: two layers, T = 203."""
    from indextts_amd.semantic import SemanticModel
    from oracle import semantic as osem
    cfg = dataclasses.replace(W2VBertConfig(), num_layers=2)
    w = weights.synth_w2vbert_weights(cfg, tag="t/w2vbert/full")
    sm = SemanticModel(w, cfg, device=device)
    B, T = 2, 203
    feats = torch.from_numpy(synth.uniform("t/w2vbert/full/feats", (B, T, cfg.input_dim), 1.5))
    lens = torch.tensor([203, 150])
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        want = osem.get_emb(tw, cfg, feats, mask).numpy()
    got = sm(feats, mask).cpu().numpy()
    scale = max(1.0, np.abs(want).max())
    for b in range(B):
        n = int(lens[b])
        err = np.abs(got[b, :n] - want[b, :n])
        assert err.max() <= 3e-4 * scale and err.mean() <= 3e-5 * scale, (b, err.max(), err.mean())
