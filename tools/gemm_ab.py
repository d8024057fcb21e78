#!/usr/bin/env python3
"""A/B/C of the LDS-DMA split-bf16 GEMM's geometries inside ONE process (interleaved rounds, per-launch HIP events of the library's
own profiler, so the activation pre-pass is not in the number): 0 = 256 x 256 tiles, one workgroup per CU; 1 = 128 x 128, three
workgroups per CU (the shipped choice).  Also measured this way and removed again (profiles/README.md "Round 3"): 128 x 128 with a
4-deep ring at two workgroups per CU (2-7 % slower than three), and 128 x 128 on v_mfma_f32_16x16x32_bf16 (3-10 % slower).  Needs a scratch build whose gemm_bf16x3_v2_forward reads IDXTTS_EXP_V2_CFG (the shipped library picks the geometry by
shape and reads no environment variable).

    python tools/gemm_ab.py [rounds] [iters]
"""
import ctypes
import os
import sys
from ctypes import c_double, c_long, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def family_ms(name):
    n = lib.idxtts_profile_num_kernels()
    for i in range(n):
        if lib.idxtts_profile_kernel_name(i).decode() == name:
            ms, fl, by, cnt = c_double(), c_double(), c_double(), c_long()
            lib.idxtts_profile_read(i, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(cnt))
            return ms.value, cnt.value
    raise KeyError(name)


def run(M, N, K, act, with_res, rounds, iters):
    g = torch.Generator().manual_seed(M + N + K)
    w = (torch.rand(N, K, generator=g) - 0.5).contiguous()
    b = (torch.rand(N, generator=g) - 0.5).contiguous()
    x = (torch.rand(M, K, generator=g) - 0.5).to(dev).contiguous()
    n_out = N // 2 if act == 3 else N
    res = (torch.rand(M, n_out, generator=g) - 0.5).to(dev).contiguous() if with_res else None
    h = c_void_p()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(w), _lib.ptr(b), N, K, 0, ctypes.byref(h)))
    st = _lib.current_stream()
    ARMS = ("0", "1")
    outs, times = {}, {a: [] for a in ARMS}

    def call(y):
        _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), n_out, _lib.ptr(res) if with_res else None, n_out if with_res else 0, M, act, 1, st))

    for arm in ARMS:
        os.environ["IDXTTS_EXP_V2_CFG"] = arm
        y = torch.full((M, n_out), float("nan"), device=dev)
        call(y)
        torch.cuda.synchronize()
        outs[arm] = y
    for _ in range(rounds):
        for arm in ARMS:
            os.environ["IDXTTS_EXP_V2_CFG"] = arm
            y = outs[arm]
            call(y)
            torch.cuda.synchronize()
            lib.idxtts_profile_enable(1)
            for _ in range(iters):
                call(y)
            torch.cuda.synchronize()
            ms, cnt = family_ms("gemm_bf16x3_v2_kernel")
            lib.idxtts_profile_enable(0)
            times[arm].append(ms / max(cnt, 1))
    os.environ.pop("IDXTTS_EXP_V2_CFG", None)
    lib.idxtts_linear_destroy(h)
    diff = max((outs["0"] - outs[a]).abs().max().item() for a in ARMS)
    finite = all(bool(torch.isfinite(outs[a]).all()) for a in ARMS)
    ref = None
    if M * N * K <= 50208 * 512 * 512:
        yr = x.double() @ w.to(dev).double().t() + b.to(dev).double()
        if act == 3:
            yr = yr.view(M, N // 64, 2, 32)
            gte, lin = yr[:, :, 0], yr[:, :, 1]
            yr = (gte * torch.sigmoid(gte) * lin).reshape(M, N // 2)
        if with_res:
            yr = yr + res.double()
        ref = max((outs[a].double() - yr).abs().max().item() for a in ARMS)
    med = {a: sorted(t)[len(t) // 2] for a, t in times.items()}
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d} act={act} res={int(with_res)}  " + "  ".join(f"cfg{a} {med[a] * 1e3:7.1f} us ({fl / med[a] / 1e9:5.1f} TF-eq)" for a in ARMS)
          + f"  max|cfg0-cfgX| {diff:.2e} finite={finite} max|x-fp64| {ref if ref is None else format(ref, '.2e')}", flush=True)


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    shapes = [(50208, 512, 512, 0, True), (50208, 1536, 512, 0, False), (50208, 512, 1536, 0, True), (50208, 3072, 512, 3, False),
              (28672, 512, 512, 0, True), (10848, 3840, 1280, 0, False), (10848, 5120, 1280, 1, False), (2066, 1536, 512, 0, False),
              (750, 1024, 1024, 0, True), (8192, 8192, 1024, 0, False), (5000, 200, 608, 0, True)]
    for (M, N, K, act, r) in shapes:
        run(M, N, K, act, r, rounds, iters)
