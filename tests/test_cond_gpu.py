"""GPU parity of the prompt-conditioning encoders through the C ABI: conformer + perceiver + emotion vector
against fixtures the REFERENCE's own UnifiedVoice produced (tests/golden/gpt_ref.npz) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import CondModuleConfig, GPTConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(golden_dir, device):
    from indextts_amd.cond import ConditioningEncoders
    g = np.load(os.path.join(golden_dir, "gpt_ref.npz"))
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="golden/gptref")
    w.update(weights.synth_gpt_cond_weights(cfg, tag="golden/gptref"))
    enc = ConditioningEncoders(w, cfg, device=device)
    spk = torch.from_numpy(synth.uniform("golden/gptref/spk", (1, 23, 1024), 1.0))
    emo = torch.from_numpy(synth.uniform("golden/gptref/emo", (1, 19, 1024), 1.0))
    return g, cfg, w, enc, spk, emo


def test_get_conditioning_vs_reference_fixture(ref):
    g, cfg, w, enc, spk, emo = ref
    ln = torch.tensor([spk.shape[-1]])                       # the reference's "length" (infer_v2.py:751): no padding
    lat = enc.get_conditioning(spk.transpose(1, 2), ln)
    np.testing.assert_allclose(lat.cpu().numpy(), g["cond_latent"], rtol=0, atol=1e-4)


def test_emovec_and_merge_vs_reference_fixture(ref):
    g, cfg, w, enc, spk, emo = ref
    ln = torch.tensor([1024])
    np.testing.assert_allclose(enc.get_emovec(spk, ln).cpu().numpy(), g["emovec_spk"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(enc.get_emovec(emo, ln).cpu().numpy(), g["emovec_emo"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(enc.merge_emovec(spk, emo, ln, ln, alpha=0.6).cpu().numpy(), g["emovec_merged"], rtol=0, atol=1e-4)


def test_ragged_prompts_vs_reference_fixture(ref):
    g, cfg, w, enc, _, _ = ref
    pair = torch.from_numpy(synth.uniform("golden/gptref/pair", (2, 21, 1024), 1.0))
    pair[1, 14:] = 0.0
    plen = torch.tensor([21, 14])
    np.testing.assert_allclose(enc.get_conditioning(pair.transpose(1, 2), plen).cpu().numpy(), g["cond_latent_ragged"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(enc.get_emovec(pair, plen).cpu().numpy(), g["emovec_ragged"], rtol=0, atol=1e-4)


def test_mid_size_encoder_vs_oracle(device):
    """Wider than the fixture (head_dim 64 and 128, several 128-row tiles, K-split embed GEMM over more tiles) vs the CPU oracle."""
    from dataclasses import replace
    from indextts_amd.cond import ConditioningEncoders
    from oracle import cond as oc
    cfg = replace(GPTConfig.tiny(), model_dim=256, cond_latents=8,
                  cond_module=CondModuleConfig(output_size=128, linear_units=256, attention_heads=2, num_blocks=2),
                  emo_cond_module=CondModuleConfig(output_size=128, linear_units=192, attention_heads=1, num_blocks=1))
    w = weights.synth_gpt_cond_weights(cfg, tag="t/cond/mid")
    enc = ConditioningEncoders(w, cfg, device=device)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    x = torch.from_numpy(synth.uniform("t/cond/mid/x", (2, 301, 1024), 1.0))
    lens = torch.tensor([301, 200])
    x[1, 200:] = 0.0
    ref_lat = oc.get_conditioning(tw, cfg, x, lens)
    ref_emo = oc.get_emovec(tw, cfg, x, lens)
    lat = enc.get_conditioning(x.transpose(1, 2), lens).cpu()
    ev = enc.get_emovec(x, lens).cpu()
    assert (lat - ref_lat).abs().max().item() <= 1e-4 * max(1.0, ref_lat.abs().max().item())
    assert (ev - ref_emo).abs().max().item() <= 1e-4 * max(1.0, ref_emo.abs().max().item())
