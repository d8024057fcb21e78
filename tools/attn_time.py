"""Per-launch time of the DiT planes attention at the bench shape (B = 32 with CFG -> 64 rows, T = 1130, prompt 689), by the library's own
per-launch HIP events: `python tools/attn_time.py` on a GPU box.  With a `-DATT_TIMING` build of csrc/attention.hip the kernel also
prints per-tile `s_memtime` stamps of one workgroup (wait + barrier / DMA issue / QK / softmax / PV)."""
import sys, torch
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
for _ in range(2): sm.estimator(x, px, lens, t, st, mu)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(3): sm.estimator(x, px, lens, t, st, mu)
torch.cuda.synchronize()
prof = _lib.profile_read(); _lib.profile_enable(False)
fa = prof["flash_attn_f32"]
print(f"flash attention: {fa['launches']} launches, {1000 * fa['ms'] / fa['launches']:.1f} us each", flush=True)
