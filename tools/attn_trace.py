"""Per-workgroup timeline of one planes-attention launch: start / end (`s_memrealtime`), XCC and CU of every workgroup, read back through
`idxtts_debug_att_trace` -- a symbol that exists only in a `-DATT_TIMING` build of csrc/attention.hip (diagnostic; the product library
does not carry it).  Prints residency per CU, workgroup durations and the dispatch waves (profiles/README.md, last kernel pass)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "index-tts_amd")
from indextts_amd import weights, _lib
from indextts_amd.config import PipelineConfig
from indextts_amd.s2mel import S2Mel
dev = torch.device("cuda", 0)
cfg = PipelineConfig()
sm = S2Mel(weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel"), cfg.s2mel, device=dev)
B, Tp, T = 32, 689, 1130
C = cfg.s2mel.in_channels
x = torch.randn(B, C, T, device=dev); px = torch.zeros(B, C, T, device=dev); px[..., :Tp] = torch.randn(B, C, Tp, device=dev)
st = torch.randn(B, cfg.s2mel.style_dim, device=dev); mu = torch.randn(B, T, cfg.s2mel.content_dim, device=dev); t = torch.full((B,), 0.4)
lens = torch.LongTensor([T] * B)
for _ in range(3): sm.estimator(x, px, lens, t, st, mu)
torch.cuda.synchronize()
lib = _lib.load()
n = 2304
buf = (ctypes.c_longlong * (4 * n))()
lib.idxtts_debug_att_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.idxtts_debug_att_trace(buf, 4 * n)
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 4)
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0; end = (a[:, 1] - t0) / 100.0      # us
hw = a[:, 2] & 0xffffffff; xcc = a[:, 2] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print("rc", rc, "kernel span us", end.max(), "wg duration us: mean %.1f min %.1f max %.1f" % ((end - start).mean(), (end - start).min(), (end - start).max()))
print("xcc ids", np.unique(xcc & 0xf, return_counts=True))
key = (xcc & 0xf) * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "WGs per CU min/max", cnt.min(), cnt.max())
for tq in (20, 100, 200, 300, 400):
    live = (start <= tq) & (end > tq)
    print("t=%d us: live WGs %d on %d CUs" % (tq, live.sum(), len(np.unique(key[live]))))
order = np.argsort(start)
print("start times (us) of every 128th WG in start order:", [round(float(start[order[i]]), 1) for i in range(0, n, 128)])
print("loop cycles mean", a[:, 3].mean())
