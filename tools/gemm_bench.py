#!/usr/bin/env python3
"""Micro-benchmark of the token-major GEMM kernels through the C ABI (idxtts_linear_fwd).
    python tools/gemm_bench.py            # a sweep of hot-path shapes, fp32 MFMA vs split-bf16
"""
import ctypes
import os
import sys
from ctypes import c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib  # noqa: E402


def bench(M, N, K, mode, iters=20):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    w = (torch.rand(N, K) - 0.5).contiguous()
    x = (torch.rand(M, K, device=dev) - 0.5).contiguous()
    y = torch.empty(M, N, device=dev)
    h = c_void_p()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(w), None, N, K, 0, ctypes.byref(h)))
    st = _lib.current_stream()
    for _ in range(3):
        _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), N, None, 0, M, 0, mode, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), N, None, 0, M, 0, mode, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    lib.idxtts_linear_destroy(h)
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    if len(sys.argv) >= 5:      # one shape, one mode (for rocprofv3 --pmc passes): M N K mode [iters]
        M, N, K, mode = (int(v) for v in sys.argv[1:5])
        r = bench(M, N, K, mode, int(sys.argv[5]) if len(sys.argv) > 5 else 5)
        print(f"M={M} N={N} K={K} mode={mode}: {r[0]:.3f} ms {r[1]:.1f} TF", flush=True)
        sys.exit(0)
    shapes = [(50208, 1536, 512), (50208, 3072, 512), (50208, 512, 1536), (50208, 512, 512), (50208, 128, 512), (50208, 512, 864),
              (50208, 1024, 2560), (13488, 3840, 1280), (13488, 1280, 5120), (8192, 8192, 1024)]
    for (M, N, K) in shapes:
        a = bench(M, N, K, 0)
        b = bench(M, N, K, 1)
        print(f"M={M:6d} N={N:5d} K={K:5d}   fp32 {a[0]:8.3f} ms {a[1]:7.1f} TF   split-bf16 {b[0]:8.3f} ms {b[1]:7.1f} TF-eq", flush=True)
