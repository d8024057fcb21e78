"""How N concurrent decode chains (gpt_stage of configs[2]: 16 utterances x 512 codes each, one stream + host thread per chain)
share the GPU when nothing else runs: wall time of N chains started together, per chain."""
import sys, threading, concurrent.futures, time
import torch
sys.path.insert(0, "index-tts_amd")
from indextts_amd import synth, weights
from indextts_amd.config import PipelineConfig
from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning

dev = torch.device("cuda", 0)
cfg = PipelineConfig()
wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
ws = weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel")
wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=dev, gpt_weight_format=sys.argv[1] if len(sys.argv) > 1 else "bf16")
tts.gpt.MAX_WORKSPACES = 16
cond = PromptConditioning.synthetic(cfg, prompt_frames=689, tag="bench/prompt").to(dev)
B, L, M = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 128, 512
from indextts_amd import _lib
if len(sys.argv) > 3 and int(sys.argv[3]):
    _lib.set_decode_plane_rows(int(sys.argv[3]))
if len(sys.argv) > 4 and sys.argv[4] == "narrow":
    _lib.load(); _lib.set_decode_geometry(True)
text = torch.from_numpy(synth.integers("bench/text/rank0", (64, L), 2, cfg.gpt.number_text_tokens))[:B]
import warnings; warnings.simplefilter("ignore")
tls = threading.local()
def job(k):
    torch.cuda.set_device(dev)
    if not hasattr(tls, "s"):
        tls.s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(tls.s):
        st = tts.gpt_stage(text, cond, max_mel_tokens=M)
        tls.s.synchronize()
    return st
for n in ((1, 2, 3, 4, 6, 8, 12) if B <= 16 else (1, 2, 3, 4)):
    with concurrent.futures.ThreadPoolExecutor(n) as pool:
        list(pool.map(job, range(n)))          # warm the lanes (workspaces, streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        list(pool.map(job, range(n)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{n:2d} chains of {B} rows together: {dt:.3f} s = {dt / n * 16 / B:.3f} s per 16 utterances", flush=True)
