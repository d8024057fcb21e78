"""Debug: do the vocoder / s2mel kernels write outside their workspace or output?  Guard bands around both."""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, "index-tts_amd")
from indextts_amd import synth, weights, _lib
from indextts_amd.config import PipelineConfig
from indextts_amd.vocoder import BigVGAN

dev = torch.device("cuda", 0)
cfg = PipelineConfig()
wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
voc = BigVGAN(wv, cfg.bigvgan)
lib = _lib.load()
B, Tm = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 880
G = 64 << 20
need = voc.workspace_bytes(B, Tm)
print("workspace bytes", need, flush=True)
buf = torch.full((G + need + G,), 0xA5, dtype=torch.uint8, device=dev)
ws = buf[G:G + need]
nw = B * Tm * 256
wbuf = torch.full((G // 4 + nw + G // 4,), 777.0, dtype=torch.float32, device=dev)
wav = wbuf[G // 4: G // 4 + nw]
mbuf = torch.full((G // 4 + B * 80 * Tm + G // 4,), 555.0, dtype=torch.float32, device=dev)
mel = mbuf[G // 4: G // 4 + B * 80 * Tm]
mel.copy_(torch.from_numpy(weights.synth_mel("t/guard/mel", B, 80, Tm)).reshape(-1).to(dev))
torch.cuda.synchronize()
for mode in (1, 0):
    _lib.set_gemm_mode(mode)
    _lib.check(lib.idxtts_bigvgan_fwd(voc._h, _lib.ptr(mel), _lib.ptr(wav), B, Tm, _lib.ptr(ws), need, 1, 0, None, _lib.current_stream()))
    torch.cuda.synchronize()
    lo, hi = buf[:G], buf[G + need:]
    print("mode", mode, "ws guard low intact", bool((lo == 0xA5).all()), "high intact", bool((hi == 0xA5).all()),
          "| wav guards", bool((wbuf[:G // 4] == 777.0).all()), bool((wbuf[G // 4 + nw:] == 777.0).all()),
          "| mel guards", bool((mbuf[:G // 4] == 555.0).all()), bool((mbuf[G // 4 + B * 80 * Tm:] == 555.0).all()), flush=True)
    for name, t, pat in (("ws-high", hi, 0xA5), ("ws-low", lo, 0xA5)):
        bad = (t != pat).nonzero().flatten()
        if len(bad):
            print("   ", name, "corrupted bytes", len(bad), "first", int(bad[0]), "last", int(bad[-1]))
    for name, t in (("wav-low", wbuf[:G // 4]), ("wav-high", wbuf[G // 4 + nw:])):
        bad = (t != 777.0).nonzero().flatten()
        if len(bad):
            print("   ", name, "corrupted floats", len(bad), "first", int(bad[0]), "last", int(bad[-1]))
_lib.set_gemm_mode(1)
