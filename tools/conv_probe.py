"""Per-launch time of the vocoder's narrow split-bf16 convolutions by tap count (is an iteration or the tile's fixed cost the bound?)."""
import sys, time, torch
sys.path.insert(0, "index-tts_amd")
from indextts_amd import _lib
from indextts_amd.vocoder import Conv1d
dev = torch.device("cuda", 0)
_lib.load(); _lib.set_gemm_mode(1)
B = 16
for C, T in ((24, 225280), (48, 112640), (96, 56320), (192, 28160)):
    x = torch.randn(B, C, T, device=dev); out = torch.empty_like(x); res = torch.randn_like(x)
    for k in (1, 3, 7, 11):
        conv = Conv1d(torch.randn(C, C, k) * 0.05, torch.zeros(C))
        for _ in range(3): conv(x, out=out, residual=res)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): conv(x, out=out, residual=res)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        gb = 3 * x.numel() * 4 / 1e9
        print(f"C={C:3d} T={T:6d} k={k:2d}: {1e6 * dt:7.1f} us  ({gb / dt / 1e3:5.2f} TB/s of x + res + y)", flush=True)
