"""CPU: the Slaney mel basis restated in indextts_amd/audio.py against transformers' port of librosa.filters.mel, and the oracle's
mel_spectrogram against the fixture the reference's own function produced (tests/golden/make_golden.py::make_melspec)."""
import os

import numpy as np
import torch

from indextts_amd.audio import slaney_mel_basis
from oracle import audio as oa

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "melspec.npz"))


def test_mel_basis_matches_third_party_port():
    mb = slaney_mel_basis(22050, 1024, 80, 0, None)
    assert mb.shape == (80, 513) and mb.dtype == np.float32
    assert np.abs(mb - G["mel_basis"]).max() <= 1e-7 * max(1.0, np.abs(G["mel_basis"]).max())
    mb8 = slaney_mel_basis(22050, 1024, 80, 0, 8000)
    assert np.all(mb8[:, 373:] == 0) and mb8[-1].argmax() < 372          # nothing above 8 kHz


def test_oracle_mel_matches_reference_fixture():
    with torch.no_grad():
        mel = oa.mel_spectrogram(torch.from_numpy(G["audio"]), torch.from_numpy(G["mel_basis"]))
    assert mel.shape == G["mel"].shape == (2, 80, (22050 + 333) // 256)
    assert np.abs(mel.numpy() - G["mel"]).max() <= 1e-5
