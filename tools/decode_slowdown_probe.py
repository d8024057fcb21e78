#!/usr/bin/env python3
"""How much slower does one decode lane run beside each kind of acoustic kernel?  A 16-utterance greedy decode (full-size GPT, bf16 weights and
cache, graph replay, high-priority stream) is timed alone and while ONE kind of kernel loops on a second stream: the split-bf16 GEMM at the
DiT's shapes (HBM-bound N = 512 and MFMA-bound N = 3072 SwiGLU), the 1536-channel vocoder convolution, the anti-alias activation.

    python tools/decode_slowdown_probe.py [tokens] [narrow 0|1]
"""
import ctypes
import os
import sys
import threading
import time
from ctypes import c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib, synth, weights  # noqa: E402
from indextts_amd.config import PipelineConfig  # noqa: E402
from indextts_amd.gpt import UnifiedVoice  # noqa: E402
from indextts_amd.vocoder import Conv1d, anti_alias_activation_forward, kaiser_sinc_filter12  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96
narrow = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
lib = _lib.load()
_lib.set_decode_geometry(narrow)
dev = torch.device("cuda", 0)
cfg = PipelineConfig()
wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
gpt = UnifiedVoice(wg, cfg.gpt, device=dev, weight_format="bf16")
B, L = 16, 128
text = torch.from_numpy(synth.integers("bench/text/rank0", (B, L), 2, cfg.gpt.number_text_tokens))
lat = torch.from_numpy(synth.uniform("dbg/lat", (B, 32, cfg.gpt.model_dim), 0.5)).to(dev)
emo = torch.from_numpy(synth.uniform("dbg/emo", (B, cfg.gpt.model_dim), 0.5)).to(dev)
lo, hi = torch.cuda.Stream.priority_range()
sd = torch.cuda.Stream(device=dev, priority=hi)


def decode():
    with torch.cuda.stream(sd):
        t0 = time.perf_counter()
        gpt.inference_speech(lat, text, emo_vec=emo, max_generate_length=M, repetition_penalty=10.0, do_sample=False, num_beams=1)
        sd.synchronize()
        return time.perf_counter() - t0


def mk_gemm(Mr, N, K, act):
    g = torch.Generator().manual_seed(N + K)
    w = (torch.rand(N, K, generator=g) - 0.5).contiguous()
    b = torch.zeros(N)
    x = (torch.rand(Mr, K, generator=g) - 0.5).to(dev).contiguous()
    n_out = N // 2 if act == 3 else N
    y = torch.empty(Mr, n_out, device=dev)
    h = c_void_p()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(w), _lib.ptr(b), N, K, 0, ctypes.byref(h)))

    def run():
        _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), n_out, None, 0, Mr, act, 1, _lib.current_stream()))
    return run


def mk_conv(C, T, k):
    conv = Conv1d(torch.randn(C, C, k) * 0.05, torch.zeros(C))
    x = torch.randn(B, C, T, device=dev)
    out = torch.empty(B, C, T, device=dev)
    return lambda: conv(x, out=out)


def mk_aa(C, T):
    f = kaiser_sinc_filter12().to(dev)
    x = torch.randn(B, C, T, device=dev)
    al, be = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    return lambda: anti_alias_activation_forward(x, f, f, al, be)


loads = {"nothing": None, "GEMM 50208 x 512 x 512 (store-bound)": mk_gemm(50208, 512, 512, 0), "GEMM 50208 x 3072 x 512 SwiGLU (MFMA-bound)": mk_gemm(50208, 3072, 512, 3),
         "Conv1d 1536 ch k3, T 880": mk_conv(1536, 880, 3), "Conv1d 384 ch k7, T 7040": mk_conv(384, 7040, 7), "anti-alias activation 768 ch, T 3520": mk_aa(768, 3520)}
decode(); decode()
base = None
for name, fn in loads.items():
    stop = threading.Event()

    def loop():
        torch.cuda.set_device(dev)
        sv = torch.cuda.Stream(device=dev, priority=lo)
        with torch.cuda.stream(sv):
            while not stop.is_set():
                for _ in range(16):
                    fn()
                sv.synchronize()
    th = None
    if fn is not None:
        th = threading.Thread(target=loop)
        th.start()
        time.sleep(0.3)
    ts = sorted(decode() for _ in range(3))
    stop.set()
    if th:
        th.join()
    t = ts[1]
    base = base or t
    print(f"decode of {M} tokens beside {name:48s}: {1e3 * t:8.1f} ms  ({1e6 * t / M:7.1f} us per token, {t / base:4.2f} x)", flush=True)
