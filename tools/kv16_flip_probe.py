#!/usr/bin/env python3
"""Where do the GPU's greedy codes leave the oracle's when the KV cache is bf16, and by how much did the oracle prefer its own token there?
The bench's CPU-leg utterance (1 x 32 text tokens, full-size GPT, bf16 weights): first differing step, the oracle's logit margin between its
choice and the GPU's at that step (a flip inside the noise bf16 rounding of keys / values adds -- one fp32 ulp of difference in a key can move
its bf16 rounding by 2^-9 relative -- shows up as a margin of the order of 1e-4 .. 1e-3 of the logit scale).

    python tools/kv16_flip_probe.py [codes] [weight_format]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from indextts_amd import synth, weights  # noqa: E402
from indextts_amd.config import PipelineConfig  # noqa: E402
from indextts_amd.gpt import UnifiedVoice  # noqa: E402
from indextts_amd.infer_v2 import PromptConditioning  # noqa: E402
from oracle import gpt as og  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96
fmt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = PipelineConfig()
g = cfg.gpt
wg = weights.synth_gpt_weights(g, tag="bench/gpt")
wg["mel_head.bias"][g.stop_mel_token] = -1e4
uv = UnifiedVoice(wg, g, device="cuda:0", weight_format=fmt, keep_effective=True)
tw = {k: torch.from_numpy(v) for k, v in (uv.effective_state_dict or wg).items()}
cond = PromptConditioning.synthetic(cfg, prompt_frames=689, tag="bench/prompt")
text = torch.from_numpy(synth.integers("bench/text", (16, 128), 2, g.number_text_tokens))[:1, :32].clone()
torch.set_num_threads(16)
conds = og.conds_latent(tw, g, cond.spk_cond_latent, cond.emo_vec)
for kv in ("bf16", "f32"):
    uv.set_kv_format(kv)
    codes, _ = uv.inference_speech(cond.spk_cond_latent, text, emo_vec=cond.emo_vec, max_generate_length=M, do_sample=False, num_beams=1, repetition_penalty=10.0)
    got = codes.cpu().numpy()[0]
    with torch.no_grad():
        ref, logits = og.generate_greedy(tw, g, conds, text, M, 10.0, return_logits=True, kv_round=kv == "bf16")
    ref = ref.numpy()[0]
    n = min(len(got), len(ref))
    diff = np.nonzero(got[:n] != ref[:n])[0]
    if len(diff) == 0:
        print(f"kv={kv}: all {n} codes equal", flush=True)
        continue
    s = int(diff[0])
    fake = og.prepare_gpt_inputs(tw, g, conds, text)[0]
    ids = torch.cat([fake, torch.from_numpy(ref[:s].astype(np.int64))[None]], dim=1)
    lg = og.repetition_penalty(ids, logits[:, s].float(), 10.0)[0]      # what the argmax sees
    top = torch.topk(lg, 3)
    print(f"kv={kv}: first difference at step {s} of {n}: oracle {int(ref[s])} gpu {int(got[s])}; oracle logits: own {lg[int(ref[s])].item():.6f} "
          f"gpu's {lg[int(got[s])].item():.6f} (margin {lg[int(ref[s])].item() - lg[int(got[s])].item():.3e}); top-3 {top.values.tolist()} ids {top.indices.tolist()}; "
          f"logit std {lg.std().item():.3f}", flush=True)
