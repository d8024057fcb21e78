// Cold-weight timing of the decode GEMV shapes (GPT-2 d = 1280) with ablation masks.  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -DGEMV_PROBE -Iindex-tts_amd/csrc tools/gemv_probe.hip index-tts_amd/csrc/gemv_fx.hip index-tts_amd/csrc/prof.hip -o /tmp/gemv_probe
#include <cstdio>
#include <vector>
#include "gemv16.h"
using namespace idxtts;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
namespace idxtts { int fail(const char* file, int line, const std::string& msg) { printf("%s:%d %s\n", file, line, msg.c_str()); return 1; } }

static int g_fmt = WFMT_F32;
static int time_it(const char* tag, int N, int K, int rows, bool ln, bool res, int dbg, float* pool, size_t pool_floats,
                   float* x, float* y, float* g) {
  const size_t wfl = gemv16_packed_floats(N, K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t off = 0; const int reps = 48;
  for (int it = -3; it < reps; ++it) {
    if (it == 0) CK(hipEventRecord(e0, 0));
    Gemv16Weights w; w.wp = pool + off; w.N = N; w.K = K; w.fmt = g_fmt; w.wscale = g;
    off += wfl; if (off + wfl > pool_floats) off = 0;
    GemvFXArgs a; a.xf = x; a.rows = rows; a.y = y; a.ldy = N; a.dbg = dbg; a.bias = g;
    if (ln) a.colsum = g;
    if (res) { a.res = y; a.y_frag = 1; }
    if (gemv_fx_forward(w, a, 0)) { printf("gemv failed\n"); return 1; }
  }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%-5s %-14s N=%5d K=%5d rows=%2d ln=%d res=%d dbg=%d : %7.2f us  %7.1f GB/s\n", g_fmt == WFMT_FP8 ? "fp8" : (g_fmt == WFMT_BF16 ? "bf16" : "f32"), tag, N, K, rows, ln, res, dbg, us,
         wfl * (double)wfmt_bytes(g_fmt) / us / 1e3);
  return 0;
}

int main() {
  const size_t pool_bytes = (size_t)2 << 30;
  float *pool, *x, *y, *g;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
  CK(hipMalloc(&x, 64 * 5120 * 4)); CK(hipMemset(x, 0, 64 * 5120 * 4));
  CK(hipMalloc(&y, 64 * 8448 * 4)); CK(hipMemset(y, 0, 64 * 8448 * 4));
  CK(hipMalloc(&g, 8448 * 4)); CK(hipMemset(g, 0, 8448 * 4));
  const size_t pf = pool_bytes / 4;
  // dbg: 1 no activation loads, 2 VALU instead of MFMA, 4 stop before the reduction / epilogue, 8 empty kernel (launch + dispatch only)
  for (int fmt : {WFMT_F32}) {
    g_fmt = fmt;
    for (int dbg : {0, 16, 32, 48}) {
      if (time_it("c_attn ln", 3840, 1280, 16, true, false, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("c_proj res", 1280, 1280, 16, false, true, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("c_fc ln", 5120, 1280, 16, true, false, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("head", 8194, 1280, 16, false, false, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("c_fc ln r1", 5120, 1280, 1, true, false, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("c_fc no-ln", 5120, 1280, 16, false, false, dbg, pool, pf, x, y, g)) return 1;
      if (time_it("c_attn no-ln", 3840, 1280, 16, false, false, dbg, pool, pf, x, y, g)) return 1;
    }
  }
  return 0;
}
