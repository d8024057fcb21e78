"""CPU: the oracle's semantic-feature model (oracle/semantic.py) against fixtures the container's own transformers
Wav2Vec2BertModel produced on the synthetic weights (tests/golden/make_golden.py::make_w2vbert)."""
import os

import numpy as np
import torch

from indextts_amd import weights
from indextts_amd.config import W2VBertConfig
from oracle import semantic as osem

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "w2vbert.npz"))


def _w(cfg):
    return {k: torch.from_numpy(v) for k, v in weights.synth_w2vbert_weights(cfg, tag="golden/w2vbert").items()}


def test_hidden_states_and_get_emb_match_transformers():
    cfg = W2VBertConfig.tiny()
    w = _w(cfg)
    feats, mask, lens = torch.from_numpy(G["feats"]), torch.from_numpy(G["mask"]), G["lens"]
    import dataclasses
    with torch.no_grad():
        h0 = osem.hidden_state(w, dataclasses.replace(cfg, num_layers=0), feats, mask)
        h1 = osem.hidden_state(w, dataclasses.replace(cfg, num_layers=1), feats, mask)
        emb = osem.get_emb(w, cfg, feats, mask)
        solo = osem.get_emb(w, cfg, feats[1:2, :22])
    for b in range(feats.shape[0]):            # valid frames (padded frames are compared too: the oracle follows HF there as well)
        n = int(lens[b])
        assert np.abs(h0[b, :n].numpy() - G["hidden0"][b, :n]).max() <= 2e-5
        assert np.abs(h1[b, :n].numpy() - G["hidden1"][b, :n]).max() <= 5e-5
        assert np.abs(emb[b, :n].numpy() - G["emb_ragged"][b, :n]).max() <= 1e-4
    assert np.abs(emb.numpy() - G["emb_ragged"]).max() <= 1e-4
    assert np.abs(solo.numpy() - G["emb_row1_alone"]).max() <= 1e-4
