"""GPU: the HIP semantic codec (`idxtts_repcodec_quantize`) against the reference-generated fixture (tiny sizes) and against the CPU
oracle at the real widths (1024 -> 384 x 12 ConvNeXt blocks -> 8-dim codes, 8192 entries)."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import RepCodecConfig

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "repcodec.npz"))


def _margin(z_e, codebook):
    """cosine-similarity gap between the best and second-best code of every frame: a frame whose gap is below the fp32 noise of the
    search may legitimately resolve differently"""
    e = torch.nn.functional.normalize(z_e.reshape(-1, z_e.shape[-1]))
    c = torch.nn.functional.normalize(codebook)
    top = (e @ c.t()).topk(2, dim=1).values
    return (top[:, 0] - top[:, 1]).reshape(z_e.shape[:-1])


def test_quantize_matches_reference_fixture(device):
    from indextts_amd.codec import SemanticCodec
    cfg = RepCodecConfig.tiny()
    w = weights.synth_repcodec_weights(cfg, tag="golden/repcodec")
    sc = SemanticCodec({"semantic_codec." + k: v for k, v in w.items()}, cfg, device=device)      # the prefixed form is accepted too
    idx, q = sc.quantize(torch.from_numpy(G["x"]))
    assert idx.dtype == torch.long and np.array_equal(idx.cpu().numpy(), G["indices"])
    assert np.abs(q.cpu().numpy() - G["quantized"]).max() <= 2e-5
    idx1, q1 = sc.quantize(torch.from_numpy(G["x"][:1]))
    assert np.array_equal(idx1.cpu().numpy(), G["indices_b1"]) and np.abs(q1.cpu().numpy() - G["quantized_b1"]).max() <= 2e-5


def test_full_size_vs_oracle(device):
    from indextts_amd.codec import SemanticCodec
    from oracle import codec as ocd
    cfg = RepCodecConfig()
    w = weights.synth_repcodec_weights(cfg, tag="t/repcodec/full")
    sc = SemanticCodec(w, cfg, device=device)
    B, T = 2, 301
    x = torch.from_numpy(synth.uniform("t/repcodec/full/x", (B, T, cfg.hidden_size), 1.0))
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        want_idx, want_q = ocd.quantize(tw, x)
        z_e = torch.nn.functional.conv1d(ocd.encoder(tw, x).transpose(1, 2), tw["quantizer.quantizers.0.in_project.weight"],
                                         tw["quantizer.quantizers.0.in_project.bias"]).transpose(1, 2)
    idx, q = sc.quantize(x)
    idx, q = idx.cpu(), q.cpu()
    same = idx == want_idx
    gap = _margin(z_e, tw["quantizer.quantizers.0.codebook.weight"])
    assert bool((same | (gap < 1e-4)).all()), "a frame with a clear nearest code resolved differently"
    assert same.float().mean().item() >= 0.99
    err = (q - want_q).abs().amax(-1)
    assert err[same].max().item() <= 5e-4, err[same].max().item()
