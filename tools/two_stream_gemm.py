#!/usr/bin/env python3
"""Do two DIFFERENT GEMMs of half the rows each, on two streams, finish sooner than the same two GEMMs at full M one after the
other?  (Phase diversity between the store-bound epilogues of one kernel and the MFMA-bound loops of the other.)"""
import ctypes
import os
import sys
from ctypes import c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
import torch  # noqa: E402
from indextts_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def mk(M, N, K, act=0):
    w = (torch.rand(N, K) - 0.5).contiguous()
    h = c_void_p()
    _lib.check(lib.idxtts_linear_create(_lib.ptr(w), None, N, K, 0, ctypes.byref(h)))
    x = (torch.rand(M, K, device=dev) - 0.5)
    y = torch.empty(M, N // 2 if act == 3 else N, device=dev)
    return h, x, y, (M, N, K, act)


def fwd(g, M, stream, row0=0):
    h, x, y, (_, N, K, act) = g
    n_out = y.shape[1]
    _lib.check(lib.idxtts_linear_fwd(h, c_void_p(x.data_ptr() + row0 * K * 4), K, c_void_p(y.data_ptr() + row0 * n_out * 4), n_out, None, 0, M, act, 1,
                                     c_void_p(stream.cuda_stream)))


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    M = 50208
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()
    chains = {
        "dit layer (qkv, wo, w13, w2)": [(1536, 512, 0), (512, 512, 0), (3072, 512, 3), (512, 1536, 0)],
        "qkv + w2": [(1536, 512, 0), (512, 1536, 0)],
    }
    for name, ops in chains.items():
        gs = [mk(M, N, K, act) for (N, K, act) in ops]

        def seq():
            for g in gs:
                fwd(g, M, main)

        def two(offset):
            # stream 0 walks the chain on rows [0, M/2), stream 1 on the other half, `offset` ops behind (wrapping around)
            ev = torch.cuda.Event()
            ev.record(main)
            s0.wait_event(ev); s1.wait_event(ev)
            n = len(gs)
            for i in range(n):
                fwd(gs[i], M // 2, s0, 0)
                fwd(gs[(i + offset) % n], M // 2, s1, M // 2)
            e0, e1 = torch.cuda.Event(), torch.cuda.Event()
            e0.record(s0); e1.record(s1)
            main.wait_event(e0); main.wait_event(e1)

        t_seq = timeit(seq)
        res = [f"sequential full-M {t_seq * 1e3:.0f} us"]
        for off in range(len(gs)):
            res.append(f"two streams offset {off}: {timeit(lambda: two(off)) * 1e3:.0f} us")
        print(name + ": " + "; ".join(res), flush=True)
