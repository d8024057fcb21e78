#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the LDS-DMA GEMM (diagnostic build of the library: tools/build_timing_lib.sh ->
tools/libidxtts_timing.so, gemm_bf16x3_v2.hip compiled with -DV2_TIMING).  Per workgroup (wave 1): cycles from start to the
first stage landed, main loop, epilogue (incl. store drain), and the in-kernel clock (s_memtime / s_memrealtime)."""
import ctypes
import os
import sys
from ctypes import c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", os.environ.get("STAMPS_LIB", "libidxtts_timing.so"))
lib = _lib.load()
dev = torch.device("cuda:0")


def run(M, N, K, act, with_res):
    g = torch.Generator().manual_seed(1)
    w = (torch.rand(N, K, generator=g) - 0.5).contiguous()
    x = (torch.rand(M, K, generator=g) - 0.5).to(dev).contiguous()
    n_out = N // 2 if act == 3 else N
    y = torch.empty(M, n_out, device=dev)
    res = (torch.rand(M, n_out, generator=g) - 0.5).to(dev).contiguous() if with_res else None
    h = c_void_p()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(w), None, N, K, 0, ctypes.byref(h)))
    st = _lib.current_stream()
    grid = 8 * ((N + 127) // 128) * (((M + 127) // 128 + 7) // 8)        # 128 x 128 tiles
    stamps = torch.zeros(grid * 8, dtype=torch.int64, device=dev)

    def call():
        _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), n_out, _lib.ptr(res) if with_res else None, n_out if with_res else 0, M, act, 1, st))

    lib.idxtts_dbg_v2_stamps(None)
    for _ in range(20):
        call()
    lib.idxtts_dbg_v2_stamps(c_void_p(stamps.data_ptr()))
    call()
    torch.cuda.synchronize()
    lib.idxtts_dbg_v2_stamps(None)
    s = stamps.view(grid, 8).cpu()
    s = s[s[:, 3] != 0]
    t0 = s[:, 0].min()
    pro, loop, epi = (s[:, 1] - s[:, 0]).float(), (s[:, 2] - s[:, 1]).float(), (s[:, 3] - s[:, 2]).float()
    issue = (s[:, 6] - s[:, 2]).float()
    clk = ((s[:, 3] - s[:, 0]).float() / (s[:, 5] - s[:, 4]).float().clamp(min=1) * 100.0)
    ns = K // 16
    starts = (s[:, 0] - t0).float()
    print(f"M={M} N={N} K={K} act={act} res={int(with_res)}: {len(s)} tiles; median cycles: prologue {pro.median():.0f}, loop {loop.median():.0f} "
          f"({loop.median() / ns:.0f}/stage), epilogue {epi.median():.0f} [stores issued after {issue.median():.0f}] (p10 {epi.quantile(0.1):.0f}, p90 {epi.quantile(0.9):.0f}); "
          f"clock {clk.median():.0f} MHz; kernel span {(s[:, 3].max() - t0).item()} cycles; tile starts p50 {starts.median():.0f} max {starts.max():.0f}", flush=True)
    lib.idxtts_linear_destroy(h)


if __name__ == "__main__":
    for sh in [(50208, 512, 512, 0, True), (50208, 1536, 512, 0, False), (50208, 3072, 512, 3, False), (50208, 512, 1536, 0, True), (8192, 8192, 1024, 0, False)]:
        run(*sh)
