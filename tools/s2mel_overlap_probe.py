#!/usr/bin/env python3
"""Time the CFM solver at configs[2] (B = 16, T = 689 + 880, 20 steps) with its two CFG halves on two streams vs one stream,
alone on the device; optional kernel-level view through the library's per-launch profiler (sequential by construction)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib, weights  # noqa: E402
from indextts_amd.config import PipelineConfig  # noqa: E402
from indextts_amd.s2mel import S2Mel  # noqa: E402

cfg = PipelineConfig()
dev = torch.device("cuda:0")
_lib.load()
m = S2Mel(weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel"), cfg.s2mel, device=dev)
B, Tp, Tg = 16, 689, 880
T = Tp + Tg
g = torch.Generator().manual_seed(0)
mu = torch.randn(B, T, 512, generator=g).to(dev)
prompt = torch.randn(B, 80, Tp, generator=g).to(dev)
style = torch.randn(B, 192, generator=g).to(dev)
z = torch.randn(B, 80, T, generator=g).to(dev)
xl = [T] * B
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run():
    return m.cfm_inference(mu, xl, prompt, style, None, steps, inference_cfg_rate=0.7, z=z)


outs = {}


def timed(label, on):
    _lib.set_s2mel_overlap(on)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs[(label, on)] = o
    print(f"{label}: overlap={on}: {dt * 1e3:.1f} ms per cfm call ({steps} steps)", flush=True)


for on in (False, True):
    timed("null stream, main thread", on)
s1 = torch.cuda.Stream()
for on in (False, True):
    with torch.cuda.stream(s1):
        timed("torch stream, main thread", on)
import threading


def worker():
    torch.cuda.set_device(dev)
    s2 = torch.cuda.Stream()
    for on in (False, True):
        with torch.cuda.stream(s2):
            timed("torch stream, worker thread", on)


th = threading.Thread(target=worker)
th.start()
th.join()
ref = outs[("null stream, main thread", False)]
print("bit-identical:", all(bool(torch.equal(ref, o)) for o in outs.values()))
