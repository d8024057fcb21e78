"""Scratch probe: plane-GEMV decode vs oracle, where do codes / logits part?  python tools/pl_debug.py fmt B heads"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "index-tts_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from indextts_amd import synth, weights
from indextts_amd.config import GPTConfig
from oracle import gpt as og
import test_gpt_gpu as T

fmt, B, heads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
cfg, uv, tw = T._pl_model(dev, fmt, heads, stop_bias=1.2)
L, NEW = 9, 30
lat = torch.from_numpy(synth.uniform("t/gpt/pl/lat", (B, cfg.cond_latents, cfg.model_dim), 0.5))
emo = torch.from_numpy(synth.uniform("t/gpt/pl/emo", (B, cfg.model_dim), 0.3))
text = torch.from_numpy(synth.integers("t/gpt/pl/text", (B, L), 2, cfg.number_text_tokens))
for b in range(B):
    text[b, L - (b % 4):] = cfg.stop_text_token
conds = og.conds_latent(tw, cfg, lat, emo)
for kv in ("bf16", "f32"):
    uv.set_kv_format(kv)
    with torch.no_grad():
        ref, rl = og.generate_greedy(tw, cfg, conds, text, NEW, 10.0, return_logits=True, kv_round=kv == "bf16")
    n = ref.shape[1]
    forced = torch.full((B, NEW), cfg.stop_mel_token, dtype=torch.long); forced[:, :n] = ref
    tf, _, lg = uv.inference_speech(lat, text, emo_vec=emo, max_generate_length=NEW, repetition_penalty=10.0, return_logits=True, forced_codes=forced)
    lg = lg.cpu()[:, :n]; tf = tf.cpu()[:, :n]
    d = (lg - rl).abs()
    print(kv, "max dlogit", d.max().item(), "mean", d.mean().item(), "logit absmax", rl.abs().max().item())
    print(" per-step max:", [round(x, 5) for x in d.amax(dim=(0, 2)).tolist()])
    mm = np.argwhere((tf != ref).numpy())
    print(" tf mismatches:", mm.tolist()[:10])
    for b, s in mm[:5]:
        sc = rl[b, s]
        top = torch.topk(sc, 3)
        print("  row", b, "step", s, "ref", int(ref[b, s]), "got", int(tf[b, s]), "top3", top.values.tolist(), top.indices.tolist(), "gpu vals", lg[b, s][top.indices].tolist())
