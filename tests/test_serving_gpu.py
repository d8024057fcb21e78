"""GPU: the batch pipeline for serving loops (indextts_amd/serving.py) -- several decode chains in flight on their own streams and
host threads, one acoustic stage behind them -- returns, for every batch, exactly what the sequential `synthesize_batch` returns."""
import pytest
import torch

from indextts_amd import synth, weights
from indextts_amd.config import PipelineConfig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes,coalesce,workers,acoal,exclusive", [(1, 1, 1, 1, False), (3, 1, 1, 1, False), (2, 3, 2, 1, False), (1, 4, 1, 1, False),
                                                                     (3, 1, 1, 2, False), (2, 2, 1, 3, False), (2, 2, 1, 1, True)])
def test_batch_pipeline_equals_sequential(device, lanes, coalesce, workers, acoal, exclusive):
    """coalesce > 1: a free lane decodes several waiting requests (different widths and row counts here) as ONE batch and hands each
    request's rows on; acoustic_coalesce > 1: a free acoustic worker takes several decoded requests as ONE s2mel + vocoder batch --
    the kernels treat rows independently, so every request still equals its own sequential call bit for bit.  exclusive: decode jobs and
    acoustic jobs take turns on the chip."""
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/serve/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="t/serve/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/serve/voc")
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=device)
    cond = PromptConditioning.synthetic(cfg, prompt_frames=40, tag="t/serve/prompt").to(device)
    nb, M = 7, 20
    Tg = int(M * cfg.code_to_frame)
    # widths 9, 9, 9, 10, 10, 11, 11: neighbours of equal width are what a coalescing lane merges (row counts differ)
    texts = [torch.from_numpy(synth.integers(f"t/serve/text{k}", (2 + k % 3, 9 + (k + 1) // 3 + (k == 6)), 2, cfg.gpt.number_text_tokens)) for k in range(nb)]
    noises = [torch.from_numpy(synth.uniform(f"t/serve/noise{k}", (t.shape[0], cfg.s2mel.in_channels, 40 + Tg), 1.0)).to(device)
              for k, t in enumerate(texts)]
    import warnings
    from indextts_amd import _lib
    # Merging requests changes the number of GEMM rows, and at this toy size that moves launches across the 256-row threshold
    # between the exact-fp32 and the split-bf16 kernel (a real utterance is hundreds of rows on its own): the merged cases run in
    # the exact mode, where kernel selection does not depend on the row count.
    mode = _lib.get_gemm_mode()
    if coalesce > 1 or acoal > 1:
        _lib.set_gemm_mode(_lib.GEMM_F32)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = [tts.synthesize_batch(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
            torch.cuda.synchronize()
            with BatchPipeline(tts, decode_lanes=lanes, coalesce=coalesce, acoustic_workers=workers, acoustic_coalesce=acoal, exclusive=exclusive) as pipe:
                futs = [pipe.submit(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
                got = [f.result() for f in futs]
    finally:
        _lib.set_gemm_mode(mode)
    for k in range(nb):
        assert len(got[k]) == len(want[k])
        for a, b in zip(got[k], want[k]):
            assert torch.equal(a, b), k


def test_batch_pipeline_propagates_errors(device):
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    cfg = PipelineConfig.tiny()
    tts = IndexTTS2.from_state_dicts(cfg, weights.synth_gpt_weights(cfg.gpt, tag="t/serve/gpt"), weights.synth_s2mel_weights(cfg.s2mel, tag="t/serve/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/serve/voc"), device=device)
    cond = PromptConditioning.synthetic(cfg, prompt_frames=40, tag="t/serve/prompt").to(device)
    bad = torch.full((1, 5), cfg.gpt.number_text_tokens + 7, dtype=torch.long)       # out-of-range token id: rejected on the host
    with BatchPipeline(tts, decode_lanes=2) as pipe:
        with pytest.raises(IndexError):
            pipe.submit(bad, cond, max_mel_tokens=4).result()


def test_fullsize_decode_beside_split_bf16_convolutions_is_bit_identical(device):
    """Regression test of the concurrency defect found in round 2: with packed-FP32 VALU instructions in the build, the decode
    GEMV's folded-LayerNorm statistics went wrong whenever a split-bf16 convolution of the vocoder ran on the same CUs (bf16
    MFMAs of another kernel on the SIMD), flipping greedy tokens at near-ties.  Full-size GPT, logits of 6 steps, bit for bit."""
    import threading
    import time
    from indextts_amd import _lib
    from indextts_amd.gpt import UnifiedVoice
    from indextts_amd.vocoder import Conv1d
    cfg = PipelineConfig()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    gpt = UnifiedVoice(wg, cfg.gpt, device=device, weight_format="bf16")
    B, L, M = 16, 64, 6
    text = torch.from_numpy(synth.integers("t/serve/fulltext", (B, L), 2, cfg.gpt.number_text_tokens))
    lat = torch.from_numpy(synth.uniform("t/serve/lat", (B, 32, cfg.gpt.model_dim), 0.5)).to(device)
    emo = torch.from_numpy(synth.uniform("t/serve/emo", (B, cfg.gpt.model_dim), 0.5)).to(device)
    s = torch.cuda.Stream(device=device)

    def decode():
        with torch.cuda.stream(s):
            _, _, logits = gpt.inference_speech(lat, text, emo_vec=emo, max_generate_length=M, repetition_penalty=10.0, do_sample=False,
                                                num_beams=1, return_logits=True)
            s.synchronize()
        return logits

    ref = decode()
    assert torch.equal(ref, decode())
    assert _lib.get_gemm_mode() == _lib.GEMM_BF16X3
    C, T = 768, 3520
    conv = Conv1d(torch.randn(C, C, 3) * 0.05, torch.zeros(C))
    x = torch.randn(B, C, T, device=device)
    out = torch.empty(B, C, T, device=device)
    stop = threading.Event()

    def load():
        torch.cuda.set_device(device)
        sv = torch.cuda.Stream(device=device)
        with torch.cuda.stream(sv):
            while not stop.is_set():
                for _ in range(8):
                    conv(x, out=out)
                sv.synchronize()

    th = threading.Thread(target=load)
    th.start()
    try:
        time.sleep(0.2)
        for _ in range(6):
            assert torch.equal(decode(), ref)
    finally:
        stop.set()
        th.join()


def test_fullsize_pipeline_three_lanes_equals_sequential(device):
    """The batch pipeline at FULL model size (where the concurrency defect of round 2 lived): three decode chains in flight beside the
    acoustic stage, 5 batches of 4 utterances x 48 codes with different texts; every waveform equals the sequential call bit for bit."""
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    cfg = PipelineConfig()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan"), device=device, gpt_weight_format="bf16")
    cond = PromptConditioning.synthetic(cfg, prompt_frames=300, tag="t/serve/fullprompt").to(device)
    nb, B, L, M = 5, 4, 40, 48
    Tg = int(M * cfg.code_to_frame)
    texts = [torch.from_numpy(synth.integers(f"t/serve/full/text{k}", (B, L), 2, cfg.gpt.number_text_tokens)) for k in range(nb)]
    noises = [torch.from_numpy(synth.uniform(f"t/serve/full/noise{k}", (B, cfg.s2mel.in_channels, 300 + Tg), 1.5)).to(device) for k in range(nb)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = [tts.synthesize_batch(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
        torch.cuda.synchronize()
        with BatchPipeline(tts, decode_lanes=3) as pipe:
            got = [f.result() for f in [pipe.submit(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]]
    for k in range(nb):
        for a, b in zip(got[k], want[k]):
            assert torch.equal(a, b), k


def test_merges_that_would_change_kernels_are_refused(device):
    """Round 3's red `test_batch_pipeline_equals_sequential[2-3-2]`: merged toy requests (38-76 prefill rows each) crossed the 256-row
    threshold between the exact-fp32 and the split-bf16 GEMM, so their latent pass ran on another kernel than their sequential calls
    (1e-4 relative in the waveform).  In the default (split-bf16) mode BatchPipeline now refuses such merges (`merge_keeps_kernels`):
    every request is decoded on its own, nothing has more rows than its request, and the results equal the sequential calls."""
    from indextts_amd import _lib
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    assert _lib.get_gemm_mode() == _lib.GEMM_BF16X3
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/serve/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="t/serve/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/serve/voc"), device=device)
    cond = PromptConditioning.synthetic(cfg, prompt_frames=40, tag="t/serve/prompt").to(device)
    nb, M = 5, 20
    Tg = int(M * cfg.code_to_frame)
    texts = [torch.from_numpy(synth.integers(f"t/serve/text{k}", (2 + k % 3, 9), 2, cfg.gpt.number_text_tokens)) for k in range(nb)]
    noises = [torch.from_numpy(synth.uniform(f"t/serve/noise{k}", (t.shape[0], cfg.s2mel.in_channels, 40 + Tg), 1.0)).to(device) for k, t in enumerate(texts)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = [tts.synthesize_batch(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
        torch.cuda.synchronize()
        with BatchPipeline(tts, decode_lanes=1, coalesce=3) as pipe:
            pipe.trace = []
            got = [f.result() for f in [pipe.submit(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]]
            rows = sorted(r for kind, _, _, r in pipe.trace if kind == "decode")
    assert rows == sorted(int(t.shape[0]) for t in texts)          # one decode per request: nothing was merged
    for k in range(nb):
        for a, b in zip(got[k], want[k]):
            assert torch.equal(a, b), k


def test_fullsize_merged_48_row_decodes_equal_sequential(device):
    """Dynamic batching where it is meant to be used: full-size model, bf16 weight streams, three 16-utterance requests decoded as ONE
    48-row batch on the plane GEMV (the plane path selected from 5 rows on, so the sequential 16-row calls run on it too), each
    request's acoustic stage on its own -- every waveform equals the sequential call bit for bit, and the trace shows the merge."""
    from indextts_amd import _lib
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    cfg = PipelineConfig()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    _lib.set_decode_plane_rows(5)
    try:
        tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel"),
                                         weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan"), device=device, gpt_weight_format="bf16")
        cond = PromptConditioning.synthetic(cfg, prompt_frames=200, tag="t/serve/fullprompt").to(device)
        nb, B, L, M = 3, 16, 24, 40
        Tg = int(M * cfg.code_to_frame)
        texts = [torch.from_numpy(synth.integers(f"t/serve/m48/text{k}", (B, L), 2, cfg.gpt.number_text_tokens)) for k in range(nb)]
        noises = [torch.from_numpy(synth.uniform(f"t/serve/m48/noise{k}", (B, cfg.s2mel.in_channels, 200 + Tg), 1.5)).to(device) for k in range(nb)]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = [tts.synthesize_batch(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
            torch.cuda.synchronize()
            with BatchPipeline(tts, decode_lanes=1, coalesce=3) as pipe:
                pipe.trace = []
                # the lane is busy with a first request while the three others queue up behind it: they are taken together
                first = pipe.submit(texts[0], cond, max_mel_tokens=M, noise=noises[0])
                futs = [pipe.submit(t, cond, max_mel_tokens=M, noise=n) for t, n in zip(texts, noises)]
                got = [f.result() for f in futs]
                first.result()
                rows = [r for kind, _, _, r in pipe.trace if kind == "decode"]
        assert max(rows) > 16, rows
        for k in range(nb):
            for a, b in zip(got[k], want[k]):
                assert torch.equal(a, b), k
    finally:
        _lib.set_decode_plane_rows(0)
