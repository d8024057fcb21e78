"""GPU parity of the whole hot path: IndexTTS2.synthesize_batch (GPT decode -> latent pass -> s2mel -> BigVGAN, one ragged
batch) against the per-utterance CPU oracle that follows infer_v2.py:732-881, at reduced width."""
import numpy as np
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import PipelineConfig

pytestmark = pytest.mark.gpu


def _build(device, eos_bias):
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/pipe/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = eos_bias
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="t/pipe/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/pipe/voc")
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=device)
    cond = PromptConditioning.synthetic(cfg, prompt_frames=11, tag="t/pipe/prompt")
    return cfg, wg, ws, wv, tts, cond


def test_batch_equals_per_utterance_reference_flow(device):
    from oracle import pipeline as op
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=6.5)
    B, L, MAXM = 4, 10, 24
    text = torch.from_numpy(synth.integers("t/pipe/text", (B, L), 2, cfg.gpt.number_text_tokens))
    lens = [10, 7, 10, 4]
    for b, n in enumerate(lens):
        text[b, n:] = cfg.gpt.stop_text_token
    Tp = cond.ref_mel.shape[-1]
    noise = torch.from_numpy(synth.uniform("t/pipe/noise", (B, cfg.s2mel.in_channels, Tp + int(MAXM * 1.72) + 2), 1.7))
    # the batch: noise is sliced to the batch's longest sequence exactly as each oracle call slices its own row
    twg = {k: torch.from_numpy(v) for k, v in wg.items()}
    tws = {k: torch.from_numpy(v) for k, v in ws.items()}
    refs = [op.synthesize_one(twg, tws, wv, cfg, text[b:b + 1, :lens[b]], cond, noise[b:b + 1], MAXM) for b in range(B)]
    Tg_max = max(int(r["mel"].shape[-1]) for r in refs)
    wavs, mid = tts.synthesize_batch(text, cond, max_mel_tokens=MAXM, noise=noise[:, :, :Tp + Tg_max].to(device),
                                     return_intermediates=True)
    code_lens = [r["code_len"] for r in refs]
    assert len(set(code_lens)) > 1, "fixture should be ragged"
    for b in range(B):
        r = refs[b]
        n = r["code_len"]
        assert mid["code_lens"][b] == n
        assert np.array_equal(mid["codes"][b, :n].cpu().numpy(), r["codes"][0].numpy())          # greedy tokens: bit-exact
        assert (mid["latent"][b, :n].cpu() - r["latent"][0]).abs().max().item() <= 1e-4
        Tg = r["mel"].shape[-1]
        mel_err = (mid["mel"][b, :, :Tg].cpu() - r["mel"][0]).abs()
        assert mel_err.mean().item() <= 1e-4 and mel_err.max().item() <= 2e-3                     # mel L1 (target 1e-3)
        w = wavs[b].cpu()
        assert w.shape == r["wav"].shape
        assert (w - r["wav"]).abs().max().item() <= 32767 * 2e-4                                  # waveform, int16 units


def test_infer_return_contract(device, tmp_path):
    from indextts_amd.infer_v2 import InferenceResult
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=6.5)
    seg = synth.integers("t/pipe/seg", (2, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1)          # greedy: the parity mode (the defaults are the reference's beam-sample, below)
    sr, wav = tts.infer(cond, seg, None, max_mel_tokens=16, **G)
    assert sr == 22050 and wav.dtype == np.int16 and wav.ndim == 2 and wav.shape[1] == 1
    res = tts.infer(cond, seg[0], None, return_audio=True, return_numpy=True, max_mel_tokens=16, **G)
    assert isinstance(res, InferenceResult) and res.sampling_rate == 22050 and res.duration_sec > 0 and res.rtf > 0
    path = tts.infer(cond, seg[0], str(tmp_path / "o.wav"), max_mel_tokens=16, **G)
    assert path.endswith("o.wav")
    import wave
    with wave.open(path) as f:
        assert f.getframerate() == 22050 and f.getnframes() == int(round(res.duration_sec * 22050))
    assert tts.infer(cond, [], None) is None
    with pytest.raises(ValueError):
        list(tts.infer(cond, seg[0], None, stream_return=True, return_audio=True))      # a generator: raises on first use, as the reference
    with pytest.raises(FileNotFoundError):              # a path is read like the reference's librosa.load would (audioio.py)
        tts.infer("examples/voice_01.wav", seg[0], None)
    with pytest.raises(NotImplementedError):            # anything else that is not a prompt
        tts.infer(object(), seg[0], None)
    with pytest.raises(TypeError):
        tts.infer(cond, seg[0], None, max_mel_tokens=16, no_such_kwarg=1)
    # the reference's sampling kwargs (num_beams=1): seeded draws make the run reproducible, and it differs from greedy
    torch.manual_seed(7)       # the CFM noise comes from the global RNG (reference: torch.randn in flow_matching.py:62)
    a = tts.infer(cond, seg[0], None, max_mel_tokens=16, do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1,
                  generator=torch.Generator().manual_seed(5))
    torch.manual_seed(7)
    b = tts.infer(cond, seg[0], None, max_mel_tokens=16, do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1,
                  generator=torch.Generator().manual_seed(5))
    assert a[0] == 22050 and np.array_equal(a[1], b[1])


def test_infer_defaults_are_the_references_beam_sample(device):
    """infer() with no generation kwargs = the reference's defaults (infer_v2.py:714-722): do_sample, num_beams=3, top_p .8,
    top_k 30, temperature .8, repetition_penalty 10, length_penalty 0.  Repeatable under torch.manual_seed, equal to the explicit
    spelling of those defaults, different from greedy, and -- with a torch generator -- equal to the CPU oracle's beam-sample."""
    from oracle import gpt as og
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=2.0)
    seg = synth.integers("t/pipe/beam", (1, 9), 2, cfg.gpt.number_text_tokens).tolist()[0]

    def run(**kw):
        torch.manual_seed(3)
        return tts.infer(cond, seg, None, max_mel_tokens=20, **kw)[1]
    a, b = run(), run()
    assert np.array_equal(a, b)
    c = run(do_sample=True, num_beams=3, top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0, length_penalty=0.0)
    assert np.array_equal(a, c)
    g = run(do_sample=False, num_beams=1)
    assert a.shape != g.shape or not np.array_equal(a, g)
    # token-level: the GPT stage under an explicit torch generator == the oracle fed the same draws
    text = torch.tensor([seg])
    NEW, nb, V = 20, 3, cfg.gpt.number_mel_codes
    gen = torch.Generator().manual_seed(99)
    noise = torch.stack([torch.empty(1, nb * V).exponential_(1, generator=gen) for _ in range(NEW)])
    codes, _ = tts.gpt.inference_speech(cond.spk_cond_latent, text, emo_vec=cond.emo_vec, max_generate_length=NEW, do_sample=True, num_beams=3,
                                        top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0, length_penalty=0.0,
                                        generator=torch.Generator().manual_seed(99))
    twg = {k: torch.from_numpy(v) for k, v in wg.items()}
    want = og.generate_beam(twg, cfg.gpt, og.conds_latent(twg, cfg.gpt, cond.spk_cond_latent, cond.emo_vec), text, NEW, noise, num_beams=nb)
    assert np.array_equal(codes.cpu().numpy(), want.numpy())


def test_infer_from_prompt_features_hoists_the_conditioning(device):
    """A PromptFeatures prompt: conformer + perceiver + merge_emovec run ONCE on the GPU (cached per prompt / emotion prompt / alpha),
    and the result equals infer() on the PromptConditioning those encoders produce."""
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning, PromptFeatures
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/pipe/gpt")
    wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag="t/pipe/gpt"))
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = 6.5
    tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="t/pipe/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/pipe/voc"), device=device)
    feats = PromptFeatures.synthetic(cfg, prompt_frames=11, feat_frames=31, tag="t/pipe/feats")
    emo = PromptFeatures.synthetic(cfg, prompt_frames=11, feat_frames=27, tag="t/pipe/emo")
    seg = synth.integers("t/pipe/fseg", (2, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=16)
    torch.manual_seed(5)
    _, a = tts.infer(feats, seg, None, emo_audio_prompt=emo, emo_alpha=0.6, **G)
    key = tts._cond_cache_key
    torch.manual_seed(5)
    _, a2 = tts.infer(feats, seg, None, emo_audio_prompt=emo, emo_alpha=0.6, **G)
    assert tts._cond_cache_key == key and np.array_equal(a, a2)
    both = PromptFeatures(feats.spk_cond_emb, feats.style, feats.prompt_condition, feats.ref_mel, emo.spk_cond_emb)
    cond = PromptConditioning.from_features(tts.gpt, both, emo_alpha=0.6)
    torch.manual_seed(5)
    _, b = tts.infer(cond, seg, None, **G)
    assert np.array_equal(a, b)
    torch.manual_seed(5)
    _, c = tts.infer(feats, seg, None, **G)                      # no emotion prompt: the speaker prompt serves (infer_v2.py:583-584)
    assert c.shape != a.shape or not np.array_equal(a, c)


def test_infer_emo_vector_mixes_the_emotion_banks(device):
    """emo_vector (infer_v2.py:586-615, 668-679, 756-757): the closest bank entries by cosine similarity to the prompt's style, weighted,
    mixed with (1 - sum(w)) of the prompt's own emotion vector; emo_alpha scales the weights (truncated to 4 decimals)."""
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning, PromptFeatures
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/pipe/gpt")
    wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag="t/pipe/gpt"))
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="t/pipe/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/pipe/voc"), device=device)
    feats = PromptFeatures.synthetic(cfg, prompt_frames=11, feat_frames=31, tag="t/pipe/feats")
    emo_num = [3, 2, 4]
    d = cfg.gpt.model_dim
    emo_matrix = torch.from_numpy(synth.uniform("t/pipe/emo_matrix", (sum(emo_num), d), 0.5))
    spk_matrix = torch.from_numpy(synth.uniform("t/pipe/spk_matrix", (sum(emo_num), cfg.s2mel.style_dim), 1.0))
    seg = synth.integers("t/pipe/eseg", (1, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=12)
    import warnings
    warnings.simplefilter("ignore")
    with pytest.raises(RuntimeError):
        tts.infer(feats, seg, None, emo_vector=[0.2, 0.1, 0.3], **G)
    tts.set_emotion_matrices(emo_matrix, spk_matrix, emo_num)
    torch.manual_seed(2)
    _, a = tts.infer(feats, seg, None, emo_vector=[0.2, 0.1, 0.3], emo_alpha=0.5, **G)
    # the same mix by hand
    w = torch.tensor([int(x * 0.5 * 10000) / 10000 for x in (0.2, 0.1, 0.3)])
    idx = [int(torch.argmax(torch.nn.functional.cosine_similarity(feats.style, m, dim=1))) for m in torch.split(spk_matrix, emo_num)]
    mat = torch.stack([m[i] for i, m in zip(idx, torch.split(emo_matrix, emo_num))])
    base = PromptConditioning.from_features(tts.gpt, feats, emo_alpha=1.0)
    emovec = (w[:, None] * mat).sum(0, keepdim=True).to(device) + (1 - w.sum()) * base.emo_vec
    cond = PromptConditioning(base.spk_cond_latent, emovec, base.style, base.prompt_condition, base.ref_mel)
    torch.manual_seed(2)
    _, b = tts.infer(cond, seg, None, **G)
    assert np.array_equal(a, b)
    torch.manual_seed(2)
    _, c = tts.infer(feats, seg, None, **G)
    assert not np.array_equal(a, c)


def test_infer_takes_a_string_with_a_tokenizer(device):
    """infer(text: str) (infer_v2.py:697-704): tokenize -> split_segments -> ids; equals the call with those id segments."""
    import os
    from indextts_amd.tokenizer import TextTokenizer
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=-1e4)
    tok = TextTokenizer(os.path.join(os.path.dirname(__file__), "golden", "tiny_bpe.model"))
    assert tok.vocab_size <= cfg.gpt.number_text_tokens
    text = "the quick brown fox, the lazy dog. speech synthesis is fun! what's next?"
    G = dict(do_sample=False, num_beams=1, max_mel_tokens=10, max_text_tokens_per_segment=12)
    import warnings
    warnings.simplefilter("ignore")
    with pytest.raises(RuntimeError):
        tts.infer(cond, text, None, **G)
    tts.tokenizer = tok
    torch.manual_seed(4)
    _, a = tts.infer(cond, text, None, **G)
    segs = [tok.convert_tokens_to_ids(s) for s in tok.split_segments(tok.tokenize(text), 12)]
    assert len(segs) >= 2
    torch.manual_seed(4)
    _, b = tts.infer(cond, segs, None, **G)
    assert np.array_equal(a, b)


def test_infer_streaming_contract(device):
    """stream_return (infer_v2.py:547-555, 874-886): a generator that yields, per segment, the segment's waveform ([1, n] float32
    on the CPU, scaled and clamped) and then the inter-segment silence -- and nothing else; joined, the chunks are the
    non-streamed result plus the trailing silence."""
    import types
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=6.5)
    seg = synth.integers("t/pipe/stream", (3, 6), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(do_sample=False, num_beams=1)
    torch.manual_seed(11)
    sr, whole = tts.infer(cond, seg, None, max_mel_tokens=16, interval_silence=100, **G)
    torch.manual_seed(11)
    gen = tts.infer(cond, seg, None, max_mel_tokens=16, interval_silence=100, stream_return=True, **G)
    assert isinstance(gen, types.GeneratorType)
    chunks = list(gen)
    assert len(chunks) == 2 * len(seg)
    n_sil = int(22050 * 100 / 1000.0)
    for i, c in enumerate(chunks):
        assert c.device.type == "cpu" and c.dtype == torch.float32 and c.ndim == 2 and c.shape[0] == 1
        if i % 2:
            assert c.shape[1] == n_sil and not c.any()
        else:
            assert c.shape[1] > 0 and c.abs().max() <= 32767.0
    joined = torch.cat(chunks[:-1], dim=1).to(torch.int16).numpy().T
    # non-streamed calls synthesise the segments as ONE ragged batch (same greedy codes, CFM noise drawn segment by segment in
    # order): equal to the segment-by-segment stream up to fp32 rounding of the batched kernels
    assert joined.shape == whole.shape
    assert np.abs(joined.astype(np.int32) - whole.astype(np.int32)).max() <= 3
    tts.segment_batch = 1                      # the reference's loop as written: bit for bit the streamed chunks
    torch.manual_seed(11)
    _, seq = tts.infer(cond, seg, None, max_mel_tokens=16, interval_silence=100, **G)
    assert np.array_equal(joined, seq)
    tts.segment_batch = 2                      # 3 segments as a batch of 2 and a batch of 1
    torch.manual_seed(11)
    _, two = tts.infer(cond, seg, None, max_mel_tokens=16, interval_silence=100, **G)
    assert two.shape == whole.shape and np.abs(two.astype(np.int32) - whole.astype(np.int32)).max() <= 3
    assert list(tts.infer(cond, [], None, stream_return=True)) == []


def test_stages_overlapped_on_two_streams_equal_the_sequential_flow(device):
    """gpt_stage / acoustic_stage are what bench.py's two-stage pipeline runs on two streams and two host threads: the decode
    of batch k + 1 beside the s2mel + vocoder of batch k must give bit for bit the waveforms of the sequential calls."""
    import concurrent.futures
    cfg, wg, ws, wv, tts, cond = _build(device, eos_bias=6.5)
    dev = torch.device(device)
    texts = [torch.from_numpy(synth.integers(f"t/pipe/ovl/{i}", (3, 7), 2, cfg.gpt.number_text_tokens)) for i in range(3)]

    def acoustic(st, seed):
        torch.manual_seed(seed)                 # the CFM noise is the only random draw (greedy decode)
        return [w.cpu() for w in tts.acoustic_stage(st)]
    want = [acoustic(tts.gpt_stage(t, cond, max_mel_tokens=14), 100 + i) for i, t in enumerate(texts)]

    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)

    def job(st, ev, seed):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            out = acoustic(st, seed)
            s2.synchronize()
        return out
    futs = []
    for i, t in enumerate(texts):
        with torch.cuda.stream(s1):
            st = tts.gpt_stage(t, cond, max_mel_tokens=14)
            ev = torch.cuda.Event()
            ev.record(s1)
        futs.append(pool.submit(job, st, ev, 100 + i))      # runs beside the next batch's gpt_stage
    got = [f.result() for f in futs]
    for w_seq, w_ovl in zip(want, got):
        assert len(w_seq) == len(w_ovl)
        for a, b in zip(w_seq, w_ovl):
            assert a.shape == b.shape and torch.equal(a, b)
