"""Synchronised stage timers of one prompt encode (15 s speaker + 15 s emotion prompt, full-size encoders): where a cache miss's time goes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "index-tts_amd")]
import argparse, numpy as np, torch
import bench
from indextts_amd import _lib, features
_lib.load()
ap = argparse.Namespace(workload="prompt", gpt_weights="f32", gpt_kv=None, codes=200, text_tokens=40, prompt_seconds=15.0, batch=0)
dev = torch.device("cuda", 0)
# rebuild what bench.build_prompt_or_infer builds, but keep the pieces
from indextts_amd import synth, weights
from indextts_amd.config import CamPPlusConfig, PipelineConfig, RepCodecConfig, W2VBertConfig
from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
from indextts_amd.prompt import PromptAudio, PromptEncoders
cfg = PipelineConfig()
wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt"); wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag="bench/gpt"))
ws = weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel"); wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
wcfg, ccfg, pcfg = W2VBertConfig(), RepCodecConfig(), CamPPlusConfig()
wc = weights.synth_repcodec_weights(ccfg, tag="bench/codec")
for k in ("codebook.weight", "out_project.weight", "out_project.bias"):
    ws[f"semantic_codec.quantizer.quantizers.0.{k}"] = wc[f"quantizer.quantizers.0.{k}"]
tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=dev)
enc = PromptEncoders(weights.synth_w2vbert_weights(wcfg, tag="bench/w2v"), wc, weights.synth_campplus_weights(pcfg, tag="bench/campplus"), tts.s2mel, device=dev,
                     w2vbert_cfg=wcfg, codec_cfg=ccfg, campplus_cfg=pcfg)
a16, a22, e16 = bench._prompt_audio("bench/p16", 16000, 15.0), bench._prompt_audio("bench/p22", 22050, 15.0), bench._prompt_audio("bench/e16", 16000, 15.0)

def T(name, fn, acc):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r

for rep in range(4):
    acc = {}
    f1 = T("host seamless x2", lambda: [features.seamless_m4t_features(a16), features.seamless_m4t_features(e16)], acc)
    embs = T("w2v-bert (batch of 2)", lambda: enc.get_emb_batch([a16, e16]), acc)      # includes the features again (subtract)
    _, S_ref = T("codec.quantize", lambda: enc.codec.quantize(embs[0]), acc)
    ref_mel = T("mel", lambda: enc.mel(torch.from_numpy(a22.reshape(1, -1)).to(dev)), acc)
    fb = T("host kaldi_fbank", lambda: features.kaldi_fbank(a16), acc)
    style = T("campplus", lambda: enc.campplus(torch.from_numpy((fb - fb.mean(0, keepdims=True))[None])), acc)
    pc = T("length_regulator", lambda: enc.s2mel.length_regulator(S_ref, ylens=torch.LongTensor([ref_mel.size(2)]), n_quantizers=3, f0=None)[0], acc)
    feats = enc.encode(PromptAudio(a16, a22), PromptAudio(e16))
    T("from_features (conformer + perceiver x2, emovec)", lambda: PromptConditioning.from_features(tts.gpt, feats, emo_alpha=0.7), acc)
    T("whole encode()", lambda: enc.encode(PromptAudio(a16, a22), PromptAudio(e16)), acc)
print({k: round(v * 1e3, 1) for k, v in acc.items()})
# per-kernel profile of from_features alone (per-launch HIP events on the launch stream)
_lib.profile_enable(True)
PromptConditioning.from_features(tts.gpt, feats, emo_alpha=0.7)
torch.cuda.synchronize()
prof = _lib.profile_read()
_lib.profile_enable(False)
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print(f"from_features kernel {k:28s} launches {v['launches']:5d}  {v['ms']:8.3f} ms  {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.2f} TF  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s")
