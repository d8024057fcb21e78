// Small HBM-bound element-wise kernels of the s2mel stage (everything GEMM-shaped is in gemm.hip).
#include "prof.h"
#include "s2mel_ops.h"

namespace idxtts {

__global__ __launch_bounds__(256) void rotary_qk_kernel(float* qkv, int M, int H, int seq_len, const float* rope) {
  // one thread per (row, head, pair) over q and k: 2*H*32 pairs per row
  const int pairs = 2 * H * 32;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * pairs) return;
  const int m = (int)(idx / pairs), r = (int)(idx - (size_t)m * pairs);
  const int i = r & 31;                       // pair index inside the head
  const int t = m % seq_len;
  float2* ptr = reinterpret_cast<float2*>(qkv + (size_t)m * 3 * H * 64) + r;   // q then k are contiguous: r < 2*H*32
  const float2 cs = reinterpret_cast<const float2*>(rope)[(size_t)t * 32 + i];
  const float2 v = *ptr;
  *ptr = float2{v.x * cs.x - v.y * cs.y, v.y * cs.x + v.x * cs.y};
}

int rotary_qk(float* qkv, int M, int H, int seq_len, const float* rope, hipStream_t st) {
  IDX_CHECK(qkv && rope && seq_len > 0, "rotary args");
  const size_t n = (size_t)M * 2 * H * 32;
  static const int cat = prof_register("rotary_qk_kernel");
  ProfScope prof(cat, st, 0.0, 16.0 * n);
  hipLaunchKernelGGL(rotary_qk_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, qkv, M, H, seq_len, rope);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void silu_kernel(float* y, const float* x, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = x[i]; y[i] = v / (1.0f + expf(-v)); }
}

int silu_rows(float* y, const float* x, size_t n, hipStream_t st) {
  if (!n) return 0;
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, y, x, n);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void cfm_pack_kernel(const CfmPackArgs p) {
  // block = (t-tile of 64 frames, stacked row n); threads sweep the 864 columns
  const int n = blockIdx.y, b = n % p.B;
  const bool is_cond = n < p.B;
  const int t0 = blockIdx.x * 64;
  const int W = p.x_only ? p.C : 2 * p.C + p.D + p.S;      // x_only: the columns that change from one Euler step to the next
  const int Tp = p.prompt_len[b];
  for (int e = threadIdx.x; e < 64 * W; e += 256) {
    const int tt = e / W, c = e - tt * W;
    const int t = t0 + tt;
    if (t >= p.T) break;
    float v;
    if (c < p.C) v = p.x[((size_t)b * p.C + c) * p.T + t];
    else if (c < 2 * p.C) v = (is_cond && t < Tp) ? p.prompt[((size_t)b * p.C + (c - p.C)) * p.Tp_max + t] : 0.0f;
    else if (c < 2 * p.C + p.D) v = is_cond ? p.cond[((size_t)b * p.T + t) * p.D + (c - 2 * p.C)] : p.cond_null[c - 2 * p.C];
    else v = is_cond ? p.style[(size_t)b * p.S + (c - 2 * p.C - p.D)] : 0.0f;
    p.x_in[((size_t)n * p.T + t) * p.ld + c] = v;
  }
}

int cfm_pack(const CfmPackArgs& a, hipStream_t st) {
  const double bytes = 4.0 * 2 * a.B * (double)a.T * (a.x_only ? a.C : 2 * a.C + a.D + a.S) * 2;
  static const int cat = prof_register("cfm_pack_kernel");
  ProfScope prof(cat, st, 0.0, bytes);
  hipLaunchKernelGGL(cfm_pack_kernel, dim3(cdiv(a.T, 64), 2 * a.B), dim3(256), 0, st, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void cfm_euler_kernel(const CfmEulerArgs p) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)p.B * p.C * p.T;
  if (idx >= total) return;
  const int t = (int)(idx % p.T);
  const int c = (int)((idx / p.T) % p.C);
  const int b = (int)(idx / ((size_t)p.T * p.C));
  if (t < p.prompt_len[b] || t < p.v_t0) { p.x[idx] = 0.0f; return; }
  const int vT = p.v_T > 0 ? p.v_T : p.T, tv = t - p.v_t0;
  const float vc = p.v[((size_t)b * vT + tv) * p.ldv + c];
  const float vn = p.v_null ? p.v_null[((size_t)b * vT + tv) * p.ldv + c] : p.v[((size_t)(p.B + b) * vT + tv) * p.ldv + c];
  const float dphi = (1.0f + p.cfg_rate) * vc - p.cfg_rate * vn;
  p.x[idx] = p.x[idx] + p.dt * dphi;
}

int cfm_euler(const CfmEulerArgs& a, hipStream_t st) {
  const size_t total = (size_t)a.B * a.C * a.T;
  static const int cat = prof_register("cfm_euler_kernel");
  ProfScope prof(cat, st, 0.0, 16.0 * total);
  hipLaunchKernelGGL(cfm_euler_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void gather_tail_rows_kernel(float* dst, int ld_dst, const float* src, int ld_src, int cols, int T, int t0, int Tn) {
  const int j = blockIdx.x, n = blockIdx.y;
  const float* s = src + ((size_t)n * T + t0 + j) * ld_src;
  float* d = dst + ((size_t)n * Tn + j) * ld_dst;
  if (((cols | ld_dst | ld_src) & 3) == 0) {
    for (int e = threadIdx.x; e < cols / 4; e += 256) reinterpret_cast<f32x4*>(d)[e] = reinterpret_cast<const f32x4*>(s)[e];
  } else {
    for (int e = threadIdx.x; e < cols; e += 256) d[e] = s[e];
  }
}

int gather_tail_rows(float* dst, int ld_dst, const float* src, int ld_src, int cols, int N, int T, int t0, hipStream_t st) {
  IDX_CHECK(dst && src && t0 >= 0 && t0 < T && cols > 0, "gather_tail_rows args");
  const int Tn = T - t0;
  static const int cat = prof_register("gather_tail_rows_kernel");
  ProfScope prof(cat, st, 0.0, 8.0 * N * (double)Tn * cols);
  hipLaunchKernelGGL(gather_tail_rows_kernel, dim3(Tn, N), dim3(256), 0, st, dst, ld_dst, src, ld_src, cols, T, t0, Tn);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void cfm_init_state_kernel(float* x, const float* z, const int* prompt_len, int B, int C, int T) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * C * T) return;
  const int t = (int)(idx % T), b = (int)(idx / ((size_t)T * C));
  x[idx] = t < prompt_len[b] ? 0.0f : z[idx];
}

int cfm_init_state(float* x, const float* z, const int* prompt_len, int B, int C, int T, hipStream_t st) {
  const size_t total = (size_t)B * C * T;
  hipLaunchKernelGGL(cfm_init_state_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, x, z, prompt_len, B, C, T);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---- GroupNorm(1) + Mish over token-major [B][T][C] with per-sequence valid length ----
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(1024) void gn_stats_kernel(const float* x, const int* row_len, int T, int C, float* stats) {
  // one 1024-thread workgroup per sequence: mean, then centred variance (two passes, as torch)
  __shared__ float red[16];
  __shared__ float bc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t n = (size_t)row_len[b] * C;
  const float* xb = x + (size_t)b * T * C;
  float s = 0.f;
  for (size_t i = tid; i < n; i += 1024) s += xb[i];
  s = wsum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) { float a = 0.f; for (int i = 0; i < 16; ++i) a += red[i]; bc = n ? a / (float)n : 0.f; }
  __syncthreads();
  const float mean = bc;
  float ss = 0.f;
  for (size_t i = tid; i < n; i += 1024) { const float c = xb[i] - mean; ss += c * c; }
  ss = wsum(ss);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) { float a = 0.f; for (int i = 0; i < 16; ++i) a += red[i]; stats[2 * b] = mean; stats[2 * b + 1] = n ? a / (float)n : 0.f; }
}

__global__ __launch_bounds__(256) void gn_apply_mish_kernel(float* y, const float* x, const float* gamma, const float* beta,
                                                            const int* row_len, const float* stats, int T, int C, float eps) {
  const int m = blockIdx.x;                 // row = (b, t)
  const int b = m / T, t = m - b * T;
  const bool valid = t < row_len[b];
  const float mean = stats[2 * b], rstd = rsqrtf(stats[2 * b + 1] + eps);
  for (int c = threadIdx.x; c < C; c += 256) {
    float o = 0.0f;
    if (valid) {
      const float v = (x[(size_t)m * C + c] - mean) * rstd * gamma[c] + beta[c];
      const float sp = v > 20.0f ? v : log1pf(expf(v));      // torch softplus threshold
      o = v * tanhf(sp);
    }
    y[(size_t)m * C + c] = o;
  }
}

int groupnorm1_mish(float* y, const float* x, const float* gamma, const float* beta, const int* row_len, int B, int T, int C,
                    float eps, float* stats, hipStream_t st) {
  static const int cat = prof_register("gn_stats_kernel + gn_apply_mish_kernel");
  ProfScope prof(cat, st, 0.0, 12.0 * B * (double)T * C);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(B), dim3(1024), 0, st, x, row_len, T, C, stats);
  hipLaunchKernelGGL(gn_apply_mish_kernel, dim3(B * T), dim3(256), 0, st, y, x, gamma, beta, row_len, stats, T, C, eps);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
