// BigVGAN tail: conv_post (Conv1d C->1, k=7, pad 3, no bias) + clamp[-1,1]
// (bigvgan.py:379, 384; use_tanh_at_final=false, use_bias_at_final=false in config.json:17-18).
// One output row, C (=24) input rows: HBM-bound streaming (C*4 bytes read per output sample),
// so this is a VALU kernel, not an MFMA one: each thread keeps 4 adjacent outputs and walks the
// channels with 16-byte loads (t-4, t, t+4 windows; the overlap is served by L1/L2).
#include "common.h"
#include "prof.h"

namespace idxtts {

struct ConvPostParams {
  const float* x;   // [B][C][T]
  const float* w;   // [C][7]
  float* y;         // [B][1][T]
  int C, T;
  int clamp;
};

__global__ __launch_bounds__(256) void conv_post_kernel(const ConvPostParams p) {
  extern __shared__ float wsm[];   // [C*7]
  for (int i = threadIdx.x; i < p.C * 7; i += 256) wsm[i] = p.w[i];
  __syncthreads();
  const int b = blockIdx.y, T = p.T;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= T) return;
  const float* xb = p.x + (size_t)b * p.C * T;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool fast = ((T & 3) == 0) && t >= 4 && t + 8 <= T;
  for (int c = 0; c < p.C; ++c) {
    const float* xr = xb + (size_t)c * T;
    float win[12];   // x[t-4 .. t+7]
    if (fast) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + t - 4);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(xr + t);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(xr + t + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { win[i] = a0[i]; win[4 + i] = a1[i]; win[8 + i] = a2[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int tt = t - 4 + i;
        win[i] = (tt >= 0 && tt < T) ? xr[tt] : 0.0f;
      }
    }
    const float* wc = wsm + c * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float wk = wc[k];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk, win[i + k + 1], acc[i]);   // x[t+i+k-3]
    }
  }
  float* y = p.y + (size_t)b * T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = acc[i];
    if (p.clamp) v = fminf(1.0f, fmaxf(-1.0f, v));
    if (t + i < T) y[t + i] = v;
  }
}

int conv_post_forward(float* y, const float* x, const float* w, int B, int C, int T, int clamp, hipStream_t stream) {
  IDX_CHECK(y && x && w, "null pointer");
  if (B == 0 || T == 0) return 0;
  ConvPostParams p{x, w, y, C, T, clamp};
  dim3 grid(cdiv(T, 1024), B);
  static const int cat = prof_register("conv_post_kernel");
  ProfScope prof(cat, stream, 14.0 * B * C * (double)T, 4.0 * B * (C + 1.0) * (double)T);
  hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), (size_t)C * 7 * sizeof(float), stream, p);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
