"""CPU: the oracle's semantic-codec `quantize` (oracle/codec.py) against fixtures the reference's own RepCodec class produced on
the synthetic weights (tests/golden/make_golden.py::make_repcodec)."""
import os

import numpy as np
import torch

from indextts_amd import weights
from indextts_amd.config import RepCodecConfig
from oracle import codec as ocd

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "repcodec.npz"))


def test_encoder_and_quantize_match_reference():
    cfg = RepCodecConfig.tiny()
    w = {k: torch.from_numpy(v) for k, v in weights.synth_repcodec_weights(cfg, tag="golden/repcodec").items()}
    x = torch.from_numpy(G["x"])
    with torch.no_grad():
        enc = ocd.encoder(w, x)
        idx, q = ocd.quantize(w, x)
        idx1, q1 = ocd.quantize(w, x[:1])
    assert np.abs(enc.numpy() - G["encoded"]).max() <= 2e-5
    assert np.array_equal(idx.numpy(), G["indices"]) and np.array_equal(idx1.numpy(), G["indices_b1"])
    assert np.abs(q.numpy() - G["quantized"]).max() <= 1e-5 and np.abs(q1.numpy() - G["quantized_b1"]).max() <= 1e-5
