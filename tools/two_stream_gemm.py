import ctypes, os, sys
from ctypes import c_void_p
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "index-tts_amd"))
import torch
from indextts_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def mk(M, N, K):
    w = (torch.rand(N, K) - 0.5).contiguous(); x = (torch.rand(M, K, device=dev) - 0.5).contiguous(); y = torch.empty(M, N, device=dev)
    h = c_void_p(); _lib.check(lib.idxtts_linear_create(_lib.ptr(w), None, N, K, 0, ctypes.byref(h))); return h, x, y
def run(h, x, y, M, N, K, st):
    _lib.check(lib.idxtts_linear_fwd(h, _lib.ptr(x), K, _lib.ptr(y), N, None, 0, M, 0, 1, c_void_p(st.cuda_stream)))
for (N, K) in [(1536, 512), (512, 512), (3072, 512), (1024, 2560)]:
    M = 50208
    A = mk(M, N, K); B1 = mk(M // 2, N, K); B2 = mk(M // 2, N, K)
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        run(*A, M, N, K, s0); run(*B1, M // 2, N, K, s1); run(*B2, M // 2, N, K, s2)
    torch.cuda.synchronize()
    it = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    for _ in range(it): run(*A, M, N, K, s0)
    e1.record(s0); torch.cuda.synchronize(); t_one = e0.elapsed_time(e1) / it
    import time
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it):
        run(*B1, M // 2, N, K, s1); run(*B2, M // 2, N, K, s2)
    torch.cuda.synchronize(); t_two = (time.perf_counter() - t) * 1e3 / it
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): run(*A, M, N, K, s0)
    torch.cuda.synchronize(); t_one_wall = (time.perf_counter() - t) * 1e3 / it
    print(f"N={N} K={K}: one GEMM M={M}: {t_one:.3f} ms (wall {t_one_wall:.3f});  two halves on two streams: {t_two:.3f} ms", flush=True)
