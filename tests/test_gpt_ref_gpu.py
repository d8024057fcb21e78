"""GPU parity of the GPT stage through the C ABI against fixtures the REFERENCE's own `UnifiedVoice` produced
(tests/golden/make_golden.py::make_gpt_ref -> gpt_ref.npz): prompt layout (a2), cached decode logits and greedy codes (a3, a5),
`inference_speech` end to end (a6) and the latent pass (a7)."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import weights
from indextts_amd.config import GPTConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(golden_dir, device):
    from indextts_amd.gpt import UnifiedVoice
    g = np.load(os.path.join(golden_dir, "gpt_ref.npz"))
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="golden/gptref")
    return g, cfg, w, UnifiedVoice(w, cfg, device=device)


def test_prepare_gpt_inputs_vs_reference(ref, device):
    g, cfg, w, uv = ref
    conds = uv.conds_latent(torch.from_numpy(g["cond_latent"]).expand(3, -1, -1), torch.from_numpy(g["emovec_merged"]).expand(3, -1))
    np.testing.assert_array_equal(conds.cpu().numpy(), g["conds"])
    fake, emb, mask = uv.prepare_gpt_inputs(conds, torch.from_numpy(g["text"]))
    np.testing.assert_array_equal(fake.numpy(), g["prep_fake"])
    np.testing.assert_array_equal(mask.numpy(), g["prep_mask"])
    np.testing.assert_array_equal(emb.cpu().numpy(), g["prep_embeds"])


def test_cached_decode_logits_and_codes_vs_reference(ref, device):
    """GPT2InferenceModel.forward through its own prepare_inputs_for_generation: per-step logits and greedy codes."""
    g, cfg, w, uv = ref
    conds = torch.from_numpy(g["conds"]).to(device)
    fake, emb, mask = uv.prepare_gpt_inputs(conds, torch.from_numpy(g["text"]))
    NEW = g["step_codes"].shape[1]
    for graph in (False, True):
        out = uv.generate(fake, max_new_tokens=NEW, stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb,
                          repetition_penalty=10.0, use_graph=graph)
        np.testing.assert_array_equal(out[:, fake.shape[1]:].cpu().numpy(), g["step_codes"])
    out, logits = uv.generate(fake, max_new_tokens=NEW, stop_tokens=[cfg.stop_mel_token], attention_mask=mask, tts_embeddings=emb,
                              repetition_penalty=10.0, return_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["step_logits"], rtol=0, atol=5e-4)


def test_inference_speech_greedy_vs_reference(ref, device):
    """UnifiedVoice.inference_speech(do_sample=False, num_beams=1, repetition_penalty=10) of the reference, end to end."""
    g, cfg, w, uv = ref
    codes, lat = uv.inference_speech(torch.from_numpy(g["cond_latent"]).expand(3, -1, -1), torch.from_numpy(g["text"]),
                                     emo_vec=torch.from_numpy(g["emovec_merged"]).expand(3, -1),
                                     max_generate_length=g["speech_greedy_codes"].shape[1], do_sample=False, num_beams=1, repetition_penalty=10.0)
    np.testing.assert_array_equal(codes.cpu().numpy(), g["speech_greedy_codes"])


def test_latent_pass_vs_reference(ref, device):
    g, cfg, w, uv = ref
    B, M = g["latent_codes"].shape
    L = g["latent_text"].shape[1]
    out = uv.forward(torch.from_numpy(g["cond_latent"]).expand(B, -1, -1), torch.from_numpy(g["latent_text"]), torch.tensor([L, L]),
                     torch.from_numpy(g["latent_codes"]), torch.tensor([M, M]), emo_vec=torch.from_numpy(g["emovec_merged"]).expand(B, -1))
    np.testing.assert_allclose(out.cpu().numpy(), g["latent"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("mode", ["det", "sample"])
@pytest.mark.parametrize("graph", [False, True])
def test_beam_search_vs_reference(ref, device, tag, mode, graph):
    """num_beams=3 -- the mode IndexTTS2.infer runs by default (infer_v2.py:714-722) -- against the reference's inference_speech
    driven through the vendored beam-search loop and the reference's own BeamSearchScorer (make_gpt_ref): deterministic beams and
    beam-sample with the stored Exp(1) draws, hypotheses that finish early included (tag b)."""
    from indextts_amd.gpt import UnifiedVoice
    g, cfg, w, _ = ref
    wb = dict(w)
    wb["mel_head.bias"] = w["mel_head.bias"].copy()
    wb["mel_head.bias"][cfg.stop_mel_token] += float(g[f"beam_{tag}_stop_bias"])
    uv = UnifiedVoice(wb, cfg, device=device)
    noise = torch.from_numpy(g[f"beam_{tag}_noise"])
    want = g[f"beam_{tag}_{mode}_codes"]
    temperature, top_k, top_p = (float(x) for x in g[f"beam_{tag}_warpers"])     # c: a wide nucleus, sampled beams != deterministic ones
    codes, _ = uv.inference_speech(torch.from_numpy(g["cond_latent"]), torch.from_numpy(g["text"]), emo_vec=torch.from_numpy(g["emovec_merged"]),
                                   max_generate_length=noise.shape[0], do_sample=(mode == "sample"), num_beams=3, top_p=top_p, top_k=int(top_k),
                                   temperature=temperature, repetition_penalty=10.0, length_penalty=0.0, exp_noise=noise, use_graph=graph)
    np.testing.assert_array_equal(codes.cpu().numpy(), want)


def test_beam_sample_top_p_without_top_k_is_refused(ref, device):
    """The nucleus cut works on the top-k survivors (at most 2048 staged on the chip): top_p < 1 with top_k off would leave the
    whole vocabulary alive and cut it at an arbitrary subset -- refused loudly, as the num_beams = 1 sampler refuses it."""
    g, cfg, w, uv = ref
    with pytest.raises(RuntimeError, match="top-p needs 0 < top_k"):
        uv.inference_speech(torch.from_numpy(g["cond_latent"]), torch.from_numpy(g["text"]), emo_vec=torch.from_numpy(g["emovec_merged"]),
                            max_generate_length=4, do_sample=True, num_beams=3, top_p=0.8, top_k=0, temperature=0.8,
                            repetition_penalty=10.0, length_penalty=0.0)
    # top_p = 1 with top_k off is the plain softmax: allowed
    uv.inference_speech(torch.from_numpy(g["cond_latent"]), torch.from_numpy(g["text"]), emo_vec=torch.from_numpy(g["emovec_merged"]),
                        max_generate_length=4, do_sample=True, num_beams=3, top_p=1.0, top_k=0, temperature=0.8,
                        repetition_penalty=10.0, length_penalty=0.0)


def test_inference_speech_from_prompt_features_vs_reference(golden_dir, device):
    """inference_speech fed the raw prompt features, as the reference's call site does (infer_v2.py:760-775): conditioning encoders
    + emotion vector + greedy decode in one call, equal to the reference's greedy codes."""
    from indextts_amd import synth
    from indextts_amd.gpt import UnifiedVoice
    g = np.load(os.path.join(golden_dir, "gpt_ref.npz"))
    cfg = GPTConfig.tiny()
    w = weights.synth_gpt_weights(cfg, tag="golden/gptref")
    w.update(weights.synth_gpt_cond_weights(cfg, tag="golden/gptref"))
    uv = UnifiedVoice(w, cfg, device=device)
    spk = torch.from_numpy(synth.uniform("golden/gptref/spk", (1, 23, 1024), 1.0))
    emo = torch.from_numpy(synth.uniform("golden/gptref/emo", (1, 19, 1024), 1.0))
    ln = torch.tensor([1024])
    emovec = uv.merge_emovec(spk, emo, ln, ln, alpha=0.6)
    codes, lat = uv.inference_speech(spk, torch.from_numpy(g["text"]), emo, cond_lengths=ln, emo_cond_lengths=ln, emo_vec=emovec,
                                     max_generate_length=g["speech_greedy_codes"].shape[1], do_sample=False, num_beams=1, repetition_penalty=10.0)
    np.testing.assert_allclose(lat.cpu().numpy(), g["speech_latent"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(codes.cpu().numpy(), g["speech_greedy_codes"])


def test_embedding_index_out_of_range_is_rejected(ref, device):
    """nn.Embedding raises IndexError for an id beyond its table (the reference's behaviour); the HIP gather must not read past it."""
    g, cfg, w, uv = ref
    conds = torch.from_numpy(g["conds"]).to(device)
    bad = torch.from_numpy(g["text"]).clone()
    bad[0, 3] = cfg.number_text_tokens + 5
    with pytest.raises(IndexError):
        uv.prepare_gpt_inputs(conds, bad)
    long_text = torch.full((1, cfg.max_text_tokens + 1), 5, dtype=torch.long)       # L + 2 > text position table
    with pytest.raises(IndexError):
        uv.prepare_gpt_inputs(conds[:1], long_text)
