"""CPU: the GPT + conditioning oracles (oracle/gpt.py, oracle/cond.py) against fixtures the REFERENCE's own
`UnifiedVoice` produced (tests/golden/make_golden.py::make_gpt_ref -> gpt_ref.npz; model_v2.py imported from /root/reference
in the build container).  These pin the wrapper semantics -- prompt layout, the 0,2,3,... mel-position quirk, the double
LayerNorm head, latent assembly, conformer / perceiver / emotion vector -- to code the reference itself executed."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import GPTConfig
from oracle import cond as oc
from oracle import gpt as og


@pytest.fixture(scope="module")
def ref(golden_dir):
    g = np.load(os.path.join(golden_dir, "gpt_ref.npz"))
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="golden/gptref")
    w.update(weights.synth_gpt_cond_weights(cfg, tag="golden/gptref"))
    w = {k: torch.from_numpy(v) for k, v in w.items()}
    spk = torch.from_numpy(synth.uniform("golden/gptref/spk", (1, 23, 1024), 1.0))
    emo = torch.from_numpy(synth.uniform("golden/gptref/emo", (1, 19, 1024), 1.0))
    return g, cfg, w, spk, emo


def test_conformer_and_perceiver_match_reference(ref):
    g, cfg, w, spk, emo = ref
    ln = torch.tensor([1024])            # infer_v2.py:751-752 passes shape[-1] of a [1,T,1024] tensor as the "length"
    enc, mask = oc.conformer_encoder(w, cfg.cond_module, "conditioning_encoder", spk, ln)
    assert mask.all() and enc.shape == g["conformer_out"].shape
    np.testing.assert_allclose(enc.numpy(), g["conformer_out"], rtol=0, atol=2e-5)
    lat = oc.get_conditioning(w, cfg, spk, ln)
    np.testing.assert_allclose(lat.numpy(), g["cond_latent"], rtol=0, atol=2e-5)
    assert np.abs(g["cond_latent"]).max() > 0.5


def test_emovec_and_merge_match_reference(ref):
    g, cfg, w, spk, emo = ref
    ln = torch.tensor([1024])
    np.testing.assert_allclose(oc.get_emovec(w, cfg, spk, ln).numpy(), g["emovec_spk"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(oc.get_emovec(w, cfg, emo, ln).numpy(), g["emovec_emo"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(oc.merge_emovec(w, cfg, spk, emo, ln, ln, alpha=0.6).numpy(), g["emovec_merged"], rtol=0, atol=2e-5)
    assert np.abs(g["emovec_spk"] - g["emovec_emo"]).max() > 1e-2


def test_ragged_prompts_match_reference(ref):
    """true lengths < T: the subsampled key mask (subsampling.py:181) and the perceiver's padded mask (model_v2.py:641)."""
    g, cfg, w, _, _ = ref
    pair = torch.from_numpy(synth.uniform("golden/gptref/pair", (2, 21, 1024), 1.0))
    pair[1, 14:] = 0.0
    plen = torch.tensor([21, 14])
    np.testing.assert_allclose(oc.get_conditioning(w, cfg, pair, plen).numpy(), g["cond_latent_ragged"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(oc.get_emovec(w, cfg, pair, plen).numpy(), g["emovec_ragged"], rtol=0, atol=2e-5)


def test_prepare_gpt_inputs_matches_reference(ref):
    g, cfg, w, _, _ = ref
    conds = og.conds_latent(w, cfg, torch.from_numpy(g["cond_latent"]).expand(3, -1, -1), torch.from_numpy(g["emovec_merged"]).expand(3, -1))
    np.testing.assert_array_equal(conds.numpy(), g["conds"])
    fake, emb, mask = og.prepare_gpt_inputs(w, cfg, conds, torch.from_numpy(g["text"]))
    np.testing.assert_array_equal(fake.numpy(), g["prep_fake"])
    np.testing.assert_array_equal(mask.numpy(), g["prep_mask"])
    np.testing.assert_array_equal(emb.numpy(), g["prep_embeds"])          # gather + one add: bit-exact


def test_cached_decode_matches_reference_forward(ref):
    """GPT2InferenceModel.forward (model_v2.py:131-225) through its own prepare_inputs_for_generation, 10 steps, 3 ragged rows."""
    g, cfg, w, _, _ = ref
    codes, logits = og.generate_greedy(w, cfg, torch.from_numpy(g["conds"]), torch.from_numpy(g["text"]),
                                       max_new_tokens=g["step_codes"].shape[1], return_logits=True)
    np.testing.assert_array_equal(codes.numpy(), g["step_codes"])
    np.testing.assert_allclose(logits.numpy(), g["step_logits"], rtol=0, atol=2e-4)
    assert np.abs(g["step_logits"]).max() > 1.0


def test_greedy_codes_match_reference_inference_speech(ref):
    """UnifiedVoice.inference_speech(do_sample=False, num_beams=1, repetition_penalty=10) end to end (model_v2.py:796-895)."""
    g, cfg, w, _, _ = ref
    np.testing.assert_array_equal(g["speech_greedy_codes"], g["step_codes"])
    np.testing.assert_array_equal(g["speech_latent"], g["cond_latent"])
    codes = og.generate_greedy(w, cfg, torch.from_numpy(g["conds"]), torch.from_numpy(g["text"]), max_new_tokens=g["speech_greedy_codes"].shape[1])
    np.testing.assert_array_equal(codes.numpy(), g["speech_greedy_codes"])


def test_latent_pass_matches_reference_forward(ref):
    """UnifiedVoice.forward (model_v2.py:673-723)."""
    g, cfg, w, _, _ = ref
    B = g["latent"].shape[0]
    lat = og.latent_forward(w, cfg, torch.from_numpy(g["cond_latent"]).expand(B, -1, -1), torch.from_numpy(g["latent_text"]),
                            torch.from_numpy(g["latent_codes"]), torch.from_numpy(g["emovec_merged"]).expand(B, -1))
    np.testing.assert_allclose(lat.numpy(), g["latent"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("mode", ["det", "sample"])
def test_beam_search_matches_reference_inference_speech(ref, tag, mode):
    """num_beams=3 (the mode infer() runs by default, infer_v2.py:714-722): the reference's inference_speech under HF generate,
    deterministic beams and beam-sample with the stored Exp(1) draws.  Case c (temperature 1.6, top_p .97) is the one whose sampled
    beams differ from its deterministic ones while hypotheses still finish early."""
    g, cfg, w, _, _ = ref
    w = dict(w)
    w["mel_head.bias"] = w["mel_head.bias"].clone()
    w["mel_head.bias"][cfg.stop_mel_token] += float(g[f"beam_{tag}_stop_bias"])
    want = g[f"beam_{tag}_{mode}_codes"]
    noise = torch.from_numpy(g[f"beam_{tag}_noise"])
    temperature, top_k, top_p = (float(x) for x in g[f"beam_{tag}_warpers"])
    got = og.generate_beam(w, cfg, torch.from_numpy(g["conds"]), torch.from_numpy(g["text"]), noise.shape[0], noise,
                           num_beams=3, do_sample=(mode == "sample"), temperature=temperature, top_k=int(top_k), top_p=top_p)
    np.testing.assert_array_equal(got.numpy(), want)
    if tag == "c" and mode == "sample":
        assert not np.array_equal(want, g["beam_c_det_codes"]) and (want[:, :-1] == cfg.stop_mel_token).any()
