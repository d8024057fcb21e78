// Stress: the decode GEMV on one stream while the split-bf16 convolution runs on another.  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -Iindex-tts_amd/csrc tools/gemv_stress.hip index-tts_amd/csrc/{gemv_fx,prof,conv1d_bf16x3}.hip -o /tmp/gemv_stress
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemv16.h"
#include "conv1d.h"
using namespace idxtts;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
namespace idxtts { int fail(const char* file, int line, const std::string& msg) { printf("%s:%d %s\n", file, line, msg.c_str()); return 1; } }

__global__ void cmp_kernel(const float* y, const float* ref, int n, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && y[i] != ref[i]) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
  const int N = 3840, K = 1280, rows = 16;
  const int ln = argc > 1 ? atoi(argv[1]) : 1, C = argc > 2 ? atoi(argv[2]) : 768, T = argc > 3 ? atoi(argv[3]) : 3520, taps = argc > 4 ? atoi(argv[4]) : 3;
  const int gemv_dbg = argc > 5 ? atoi(argv[5]) : 0;
  const size_t wfl = gemv16_packed_floats(N, K);
  std::vector<float> hw(wfl), hx((size_t)64 * K), hg(N), hb(N);
  srand(1);
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
  for (auto& v : hg) v = (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
  float *w, *x, *y, *ref, *g, *b; unsigned* bad;
  CK(hipMalloc(&w, wfl * 4)); CK(hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&g, N * 4)); CK(hipMemcpy(g, hg.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&b, N * 4)); CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&y, (size_t)rows * N * 4)); CK(hipMalloc(&ref, (size_t)rows * N * 4));
  CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
  // the hog: Conv1d(C -> C, taps) over [16][C][T]
  const int Bc = 16;
  ConvWeights cw; cw.M = C; cw.Cin = C; cw.K = taps; cw.nchunk = (C + 15) / 16; cw.ups = 1;
  const size_t wbytes = (size_t)((C + 31) / 32) * cw.nchunk * taps * 2048;
  void* cwp; CK(hipMalloc(&cwp, wbytes)); CK(hipMemset(cwp, 0x3c, wbytes));      // bf16 0x3c3c = 0.0115
  cw.wp16 = cwp;
  float *cx, *cy; CK(hipMalloc(&cx, (size_t)Bc * C * T * 4)); CK(hipMalloc(&cy, (size_t)Bc * C * T * 4));
  CK(hipMemset(cx, 0, (size_t)Bc * C * T * 4));
  ConvArgs ca; ca.x = cx; ca.y = cy; ca.B = Bc; ca.T = T; ca.dil = 1; ca.pad_left = (taps - 1) / 2;
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  Gemv16Weights W; W.wp = w; W.N = N; W.K = K; W.fmt = WFMT_F32; W.wscale = nullptr;
  GemvFXArgs a; a.xf = x; a.rows = rows; a.y = y; a.ldy = N; a.dbg = gemv_dbg; a.bias = b;
  if (ln) a.colsum = g;
  if (gemv_fx_forward(W, a, sa)) return 1;
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(ref, y, (size_t)rows * N * 4, hipMemcpyDeviceToDevice));
  for (int phase = 0; phase < 2; ++phase) {
    CK(hipMemset(bad, 0, 8));
    const int iters = 2000;
    if (phase == 1)
      for (int i = 0; i < 150; ++i) if (conv1d_bf16x3_forward(cw, ca, sb)) return 1;
    for (int it = 0; it < iters; ++it) {
      CK(hipMemsetAsync(y, 0, (size_t)rows * N * 4, sa));
      if (gemv_fx_forward(W, a, sa)) return 1;
      hipLaunchKernelGGL(cmp_kernel, dim3((rows * N + 255) / 256), dim3(256), 0, sa, y, ref, rows * N, bad);
    }
    CK(hipStreamSynchronize(sa));
    const bool running = phase == 1 && hipStreamQuery(sb) == hipErrorNotReady;
    CK(hipStreamSynchronize(sb));
    unsigned hb2[2]; CK(hipMemcpy(hb2, bad, 8, hipMemcpyDeviceToHost));
    printf("gemv ln=%d dbg=%d | conv C=%d T=%d taps=%d %s: %d launches, mismatching elements %u%s\n", ln, gemv_dbg, C, T, taps, phase ? "LOADED" : "quiet ", iters, hb2[0],
           phase ? (running ? " (conv covered the whole phase)" : " (conv finished early)") : "");
  }
  return 0;
}
