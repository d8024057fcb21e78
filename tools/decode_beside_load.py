"""Which co-running vocoder kernel perturbs the decode (the investigation behind the NOPK build flag, csrc/Makefile): full-size
GPT decode of a few steps, logits compared bit for bit with the quiet run, beside loops of the fused activation and of convolutions
in both arithmetic modes.  With packed-FP32 instructions in the build, every split-bf16 convolution load changed the logits."""
import sys, threading, time
import numpy as np, torch
sys.path.insert(0, "index-tts_amd")
from indextts_amd import synth, weights, _lib
from indextts_amd.config import PipelineConfig
from indextts_amd.gpt import UnifiedVoice
from indextts_amd.vocoder import Conv1d, anti_alias_activation_forward, kaiser_sinc_filter12

dev = torch.device("cuda", 0)
cfg = PipelineConfig()
wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
gpt = UnifiedVoice(wg, cfg.gpt, device=dev, weight_format="bf16")
B, L, M = 16, 128, 8
text = torch.from_numpy(synth.integers("bench/text/rank0", (B, L), 2, cfg.gpt.number_text_tokens))
lat = torch.from_numpy(synth.uniform("dbg/lat", (B, 32, cfg.gpt.model_dim), 0.5)).to(dev)
emo = torch.from_numpy(synth.uniform("dbg/emo", (B, cfg.gpt.model_dim), 0.5)).to(dev)
s = torch.cuda.Stream(device=dev)
def decode():
    with torch.cuda.stream(s):
        codes, _, logits = gpt.inference_speech(lat, text, emo_vec=emo, max_generate_length=M, repetition_penalty=10.0, do_sample=False, num_beams=1, return_logits=True)
        s.synchronize()
    return logits
ref = decode()
assert torch.equal(ref, decode())
f = kaiser_sinc_filter12().to(dev)
def mk_aa(C, T):
    x = torch.randn(B, C, T, device=dev); al = torch.zeros(C, device=dev); be = torch.zeros(C, device=dev)
    return lambda: anti_alias_activation_forward(x, f, f, al, be)
def mk_conv(C, T, k, dil):
    w = torch.randn(C, C, k) * 0.05
    conv = Conv1d(w, torch.zeros(C))
    x = torch.randn(B, C, T, device=dev); out = torch.empty(B, C, T, device=dev)
    return lambda: conv(x, dilation=dil, out=out)
def mk_up(Cin, Cout, T, u, k):
    w = torch.randn(Cin, Cout, k) * 0.05
    conv = Conv1d(w, torch.zeros(Cout), transposed_stride=u)
    x = torch.randn(B, Cin, T, device=dev)
    return lambda: conv(x)
loads = {
    "aa_act C=768 T=3520": lambda: mk_aa(768, 3520),
    "aa_act C=24 T=225280": lambda: mk_aa(24, 225280),
    "conv C=768 k3 T=3520": lambda: mk_conv(768, 3520, 3, 1),
    "conv C=768 k11 d5 T=3520": lambda: mk_conv(768, 3520, 11, 5),
    "conv C=192 k7 T=28160": lambda: mk_conv(192, 28160, 7, 3),
    "conv C=48 k3 T=112640": lambda: mk_conv(48, 112640, 3, 1),
    "conv C=24 k11 T=225280": lambda: mk_conv(24, 225280, 11, 1),
    "up 1536->768 x4 T=880": lambda: mk_up(1536, 768, 880, 4, 8),
}
for mode in (1, 0):
    _lib.set_gemm_mode(mode)
    for name, mk in loads.items():
        fn = mk()
        stop = threading.Event()
        def run():
            torch.cuda.set_device(dev)
            sv = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(sv):
                while not stop.is_set():
                    for _ in range(8): fn()
                    sv.synchronize()
        th = threading.Thread(target=run); th.start(); time.sleep(0.2)
        bad = 0; n = 12
        for _ in range(n):
            bad += int(not torch.equal(decode(), ref))
        stop.set(); th.join()
        print(f"gemm mode {mode} load [{name}]: {bad} of {n} decodes differ", flush=True)
_lib.set_gemm_mode(1)
