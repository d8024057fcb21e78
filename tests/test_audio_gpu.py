"""GPU: the HIP log-mel spectrogram (`idxtts_melspec_forward`: STFT as a GEMM) against the reference-generated fixture and against the
CPU oracle on 15 s of audio (the prompt length the reference cuts to, infer_v2.py:628)."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "melspec.npz"))


def test_mel_matches_reference_fixture(device):
    from indextts_amd.audio import MelSpectrogram
    mf = MelSpectrogram(device=device)
    got = mf(torch.from_numpy(G["audio"]).to(device)).cpu().numpy()
    assert got.shape == G["mel"].shape
    err = np.abs(got - G["mel"])
    assert err.max() <= 2e-3 and err.mean() <= 2e-5, (err.max(), err.mean())      # log of magnitudes down to 1e-5: fp32 STFT noise near the floor


def test_fifteen_seconds_vs_oracle(device):
    from indextts_amd.audio import MelSpectrogram, slaney_mel_basis
    from oracle import audio as oa
    n = 15 * 22050
    t = np.arange(n) / 22050.0
    y = (0.5 * np.sin(2 * np.pi * (200 + 30 * np.sin(2 * np.pi * 0.7 * t)) * t) + 0.1 * synth.uniform("t/mel/noise", (n,), 1.0)).astype(np.float32)[None]
    mf = MelSpectrogram(device=device)
    got = mf(torch.from_numpy(y).to(device)).cpu()
    with torch.no_grad():
        want = oa.mel_spectrogram(torch.from_numpy(y), torch.from_numpy(slaney_mel_basis(22050, 1024, 80)))
    assert got.shape == want.shape == (1, 80, n // 256)
    err = (got - want).abs()
    assert err.max().item() <= 1e-3 and err.mean().item() <= 1e-5, (err.max().item(), err.mean().item())
    with pytest.raises(ValueError):
        mf(torch.zeros(1, 100, device=device))
