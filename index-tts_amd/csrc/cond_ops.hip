// Kernels of the prompt-conditioning encoders that are not GEMMs or LayerNorms (see cond_ops.h for the contracts).
// Reference: indextts/gpt/conformer/subsampling.py:131-181, conformer/attention.py:164-312,
// indextts/gpt/conformer_encoder.py:57-164, indextts/gpt/perceiver.py:150-177, 233-317.
// Everything here runs once per prompt on sequences of a few hundred frames: exact fp32, wave64 shuffle reductions,
// coalesced row accesses; the heavy part of the stage (the K = 261 632 input projection) is the split-K GEMM in gemm.hip.
#include <cmath>

#include "cond_ops.h"
#include "prof.h"

namespace idxtts {

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float bsum256(float v, float* red) {   // 256 threads; red: 4 floats of LDS
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// One workgroup per output frame (b, t2): the three input rows and all C filters sit in LDS; a thread owns output
// features f2 = tid, tid + 256, ... keeps their 3x3 input patch in registers and walks the channels, so every global
// store is a contiguous run over f2 and every filter read is an LDS broadcast.
__global__ __launch_bounds__(256) void sub2_conv_relu_kernel(float* a, const float* x, const float* w, const float* bias, int T, int F,
                                                             int C, int T2, int F2) {
  extern __shared__ float sm[];
  float* xs = sm;                 // [3][F]
  float* ws = sm + 3 * F;         // [C][10]: 9 taps + bias
  const int t2 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* xin = x + ((size_t)b * T + 2 * t2) * F;
  for (int e = tid; e < 3 * F; e += 256) xs[e] = xin[e];
  for (int e = tid; e < C * 10; e += 256) { const int c = e / 10, r = e - c * 10; ws[e] = r < 9 ? w[c * 9 + r] : bias[c]; }
  __syncthreads();
  float* out = a + ((size_t)b * T2 + t2) * ((size_t)C * F2);
  for (int f0 = tid; f0 < F2; f0 += 256) {
    float p[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) p[i * 3 + j] = xs[i * F + 2 * f0 + j];
    for (int c = 0; c < C; ++c) {
      const float* wc = ws + c * 10;
      float acc = wc[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) acc = fmaf(wc[r], p[r], acc);
      out[(size_t)c * F2 + f0] = fmaxf(acc, 0.0f);
    }
  }
}

int sub2_conv_relu(float* a, const float* x, const float* w, const float* bias, int B, int T, int F, int C, hipStream_t st) {
  IDX_CHECK(a && x && w && bias, "null pointer");
  IDX_CHECK(B > 0 && T >= 3 && F >= 3 && C > 0, "Conv2dSubsampling2 needs at least 3 frames and 3 features");
  const int T2 = (T - 3) / 2 + 1, F2 = (F - 3) / 2 + 1;
  const size_t lds = (size_t)(3 * F + 10 * C) * sizeof(float);
  IDX_CHECK(lds <= 64 * 1024, "input row / filter bank too large for LDS");
  static const int cat = prof_register("sub2_conv_relu_kernel");
  ProfScope prof(cat, st, 18.0 * B * T2 * (double)C * F2, 4.0 * ((double)B * T * F + (double)B * T2 * C * F2));
  hipLaunchKernelGGL(sub2_conv_relu_kernel, dim3(T2, B), dim3(256), lds, st, a, x, w, bias, T, F, C, T2, F2);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Four queries of one (batch, head) per workgroup, one wave each.  Pass 1: lanes = keys, each lane one dot product of
// length dk (two with the position term) against the query held in LDS; pass 2: wave max / sum; pass 3: lanes = output
// features, probabilities re-read from LDS (broadcast), V rows read coalesced.
__global__ __launch_bounds__(256) void seq_attn_kernel(const SeqAttnArgs p) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 4 + wave;
  const int dk = p.dk, Sk = p.Sk;
  const int nrel = p.rel_key ? p.rel_left + p.rel_right + 1 : 0, nrel4 = (nrel + 3) & ~3;
  float* qu = sm + wave * (2 * dk + nrel4 + ((Sk + 3) & ~3));      // 16-byte aligned per-wave regions
  float* qv = qu + dk;
  float* rel = qv + dk;             // [nrel] q . rel_key[d]
  float* sc = rel + nrel4;
  if (qi >= p.Sq) return;           // whole wave; no workgroup barrier below
  const float* q = p.q + (size_t)b * p.q_bs + (size_t)qi * p.ldq + h * dk;
  for (int e = lane; e < dk; e += 64) {
    const float v = q[e];
    qu[e] = v + (p.bias_u ? p.bias_u[h * dk + e] : 0.0f);
    qv[e] = v + (p.bias_v ? p.bias_v[h * dk + e] : 0.0f);
  }
  __builtin_amdgcn_wave_barrier();
  for (int r = lane; r < nrel; r += 64) {
    const float* er = p.rel_key + (size_t)r * dk;
    float s = 0.0f;
    for (int e = 0; e < dk; ++e) s = fmaf(qu[e], er[e], s);
    rel[r] = s;
  }
  __builtin_amdgcn_wave_barrier();
  const int kend = p.kend ? min(p.kend[b], Sk) : Sk;
  float mx = -INFINITY;
  for (int j = lane; j < kend; j += 64) {
    const f32x4* kr = reinterpret_cast<const f32x4*>(p.k + (size_t)b * p.k_bs + (size_t)j * p.ldk + h * dk);
    float s = 0.0f;
    for (int e = 0; e < dk / 4; ++e) {
      const f32x4 kv = kr[e];
      const f32x4 qq = *reinterpret_cast<const f32x4*>(qu + 4 * e);
      s = fmaf(qq[0], kv[0], s); s = fmaf(qq[1], kv[1], s); s = fmaf(qq[2], kv[2], s); s = fmaf(qq[3], kv[3], s);
    }
    if (p.pos) {
      const f32x4* pr = reinterpret_cast<const f32x4*>(p.pos + (size_t)j * p.ldp + h * dk);
      float s2 = 0.0f;
      for (int e = 0; e < dk / 4; ++e) {
        const f32x4 pv = pr[e];
        const f32x4 qq = *reinterpret_cast<const f32x4*>(qv + 4 * e);
        s2 = fmaf(qq[0], pv[0], s2); s2 = fmaf(qq[1], pv[1], s2); s2 = fmaf(qq[2], pv[2], s2); s2 = fmaf(qq[3], pv[3], s2);
      }
      s += s2;
    }
    if (nrel) s += rel[min(max(j - qi, -p.rel_left), p.rel_right) + p.rel_left];
    s *= p.scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wmax(mx);
  float sum = 0.0f;
  for (int j = lane; j < kend; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wsum(sum);
  __builtin_amdgcn_wave_barrier();
  const float inv = kend > 0 ? 1.0f / sum : 0.0f;
  float* o = p.o + (size_t)b * p.o_bs + (size_t)qi * p.ldo + h * dk;
  for (int d0 = lane; d0 < dk; d0 += 64) {
    const float* vr = p.v + (size_t)b * p.v_bs + h * dk + d0;
    float acc = 0.0f;
    for (int j = 0; j < kend; ++j) acc = fmaf(sc[j], vr[(size_t)j * p.ldv], acc);
    o[d0] = acc * inv;
  }
}

int seq_attn_forward(const SeqAttnArgs& a, hipStream_t st) {
  IDX_CHECK(a.q && a.k && a.v && a.o, "null pointer");
  IDX_CHECK(a.B > 0 && a.H > 0 && a.Sq > 0 && a.Sk > 0, "shape");
  IDX_CHECK(a.dk > 0 && a.dk <= 128 && (a.dk & 3) == 0, "head_dim must be a multiple of 4, at most 128");
  IDX_CHECK((a.ldk & 3) == 0 && (a.k_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.k) & 15) == 0, "k rows must be 16-byte aligned");
  if (a.pos) IDX_CHECK((a.ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(a.pos) & 15) == 0 && a.bias_u && a.bias_v, "position term");
  if (a.rel_key) IDX_CHECK(a.rel_left >= 0 && a.rel_right >= 0 && !a.bias_u && !a.pos && a.Sq == a.Sk, "relative_key: self-attention without the rel-pos term");
  const int nrel4 = a.rel_key ? (a.rel_left + a.rel_right + 1 + 3) & ~3 : 0;
  const size_t lds = (size_t)4 * (2 * a.dk + nrel4 + ((a.Sk + 3) & ~3)) * sizeof(float);
  IDX_CHECK(lds <= 64 * 1024, "key sequence too long for the short-sequence attention kernel");
  static const int cat = prof_register("seq_attn_kernel");
  ProfScope prof(cat, st, (a.pos ? 6.0 : 4.0) * a.B * a.H * (double)a.Sq * a.Sk * a.dk,
                 4.0 * a.B * a.H * a.dk * (2.0 * a.Sq + 2.0 * a.Sk));
  hipLaunchKernelGGL(seq_attn_kernel, dim3(cdiv(a.Sq, 4), a.H, a.B), dim3(256), lds, st, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_rows_kernel(float* x, int M, int d, int T, const int* len) {
  const int m = blockIdx.x;
  const int b = m / T;
  if ((m - b * T) < len[b]) return;
  for (int e = threadIdx.x; e < d; e += 256) x[(size_t)m * d + e] = 0.0f;
}

int mask_rows(float* x, int M, int d, int T, const int* len, hipStream_t st) {
  IDX_CHECK(x && len && M > 0 && d > 0 && T > 0, "mask_rows args");
  hipLaunchKernelGGL(mask_rows_kernel, dim3(M), dim3(256), 0, st, x, M, d, T, len);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
constexpr int DW_MAX_PER_THREAD = 4;    // D <= 1024

// GLU: the input rows are [2D] = (a | gate) and the conv runs over a * sigmoid(gate); SILU: swish after the LayerNorm.
// a * sigmoid(gate) of every [a | gate] row, written over a: ONE sigmoid per element (inside the depthwise loop it was evaluated once per
// tap -- 31 times at w2v-bert's kernel size)
__global__ __launch_bounds__(256) void glu_inplace_kernel(float* pw, int D, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t m = i / D;
  const int c = (int)(i - m * D);
  float* row = pw + m * 2 * (size_t)D;
  row[c] *= 1.0f / (1.0f + expf(-row[D + c]));
}

template <bool GLU, bool SILU>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(float* y, const float* pw, const float* wdw, const float* bdw, const float* gamma,
                                                        const float* beta, int T, int D, int k, int pad, float eps, int ld) {
  __shared__ float red[4];
  const int m = blockIdx.x, b = m / T, t = m - b * T, tid = threadIdx.x;
  float v[DW_MAX_PER_THREAD];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < DW_MAX_PER_THREAD; ++i) {
    const int c = tid + 256 * i;
    float acc = 0.0f;
    if (c < D) {
      acc = bdw ? bdw[c] : 0.0f;
      for (int kk = 0; kk < k; ++kk) {
        const int tt = t + kk - pad;
        if (tt < 0 || tt >= T) continue;
        const float* row = pw + ((size_t)b * T + tt) * ld;
        float a = row[c];
        if (GLU) a *= 1.0f / (1.0f + expf(-row[D + c]));
        acc = fmaf(wdw[c * k + kk], a, acc);
      }
      s += acc;
    }
    v[i] = acc;
  }
  const float mean = bsum256(s, red) / D;
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < DW_MAX_PER_THREAD; ++i) { const int c = tid + 256 * i; if (c < D) { const float dlt = v[i] - mean; ss += dlt * dlt; } }
  const float rstd = rsqrtf(bsum256(ss, red) / D + eps);
#pragma unroll
  for (int i = 0; i < DW_MAX_PER_THREAD; ++i) {
    const int c = tid + 256 * i;
    if (c < D) {
      const float n = (v[i] - mean) * rstd * gamma[c] + beta[c];
      y[(size_t)m * D + c] = SILU ? n / (1.0f + expf(-n)) : n;
    }
  }
}

int glu_dwconv_ln_silu(float* y, float* pw, const float* wdw, const float* bdw, const float* gamma, const float* beta, int B, int T,
                       int D, int k, hipStream_t st, int pad_left) {
  IDX_CHECK(y && pw && wdw && gamma && beta, "null pointer");
  IDX_CHECK(B > 0 && T > 0 && D > 0 && D <= 256 * DW_MAX_PER_THREAD && (k & 1) == 1 && pad_left < k, "shape");
  static const int cat = prof_register("dwconv_ln_kernel<true, true>");
  ProfScope prof(cat, st, 0.0, 4.0 * B * T * 3.0 * D);
  // the gated rows first, in place over the a half of pw (pw is this op's scratch input: consumed), then the depthwise taps over plain rows
  const size_t n = (size_t)B * T * D;
  hipLaunchKernelGGL(glu_inplace_kernel, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, st, pw, D, n);
  IDX_LAUNCH_CHECK();
  hipLaunchKernelGGL((dwconv_ln_kernel<false, true>), dim3(B * T), dim3(256), 0, st, y, pw, wdw, bdw, gamma, beta, T, D, k,
                     pad_left < 0 ? (k - 1) / 2 : pad_left, 1e-5f, 2 * D);
  IDX_LAUNCH_CHECK();
  return 0;
}

int dwconv_ln(float* y, const float* x, const float* wdw, const float* bdw, const float* gamma, const float* beta, int B, int T, int D, int k,
              float eps, hipStream_t st) {
  IDX_CHECK(y && x && wdw && gamma && beta, "null pointer");
  IDX_CHECK(B > 0 && T > 0 && D > 0 && D <= 256 * DW_MAX_PER_THREAD && (k & 1) == 1, "shape");
  static const int cat = prof_register("dwconv_ln_kernel<false, false>");
  ProfScope prof(cat, st, 0.0, 4.0 * B * T * 2.0 * D);
  hipLaunchKernelGGL((dwconv_ln_kernel<false, false>), dim3(B * T), dim3(256), 0, st, y, x, wdw, bdw, gamma, beta, T, D, k, (k - 1) / 2, eps, D);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_latents_ctx_kernel(float* ctx, const float* lat, const float* x, int n, int T, int d) {
  const int r = blockIdx.x, b = blockIdx.y;
  const float* src = r < n ? lat + ((size_t)b * n + r) * d : x + ((size_t)b * T + (r - n)) * d;
  float* dst = ctx + ((size_t)b * (n + T) + r) * d;
  for (int e = threadIdx.x; e < d; e += 256) dst[e] = src[e];
}

int concat_latents_ctx(float* ctx, const float* lat, const float* x, int B, int n, int T, int d, hipStream_t st) {
  IDX_CHECK(ctx && lat && x && B > 0 && n > 0 && T > 0 && d > 0, "concat args");
  hipLaunchKernelGGL(concat_latents_ctx_kernel, dim3(n + T, B), dim3(256), 0, st, ctx, lat, x, n, T, d);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void geglu_kernel(float* y, int ldy, const float* in, int F) {
  const int m = blockIdx.x;
  const float* row = in + (size_t)m * 2 * F;
  for (int j = threadIdx.x; j < ldy; j += 256) {
    float o = 0.0f;
    if (j < F) {
      const float g = row[F + j];
      o = 0.5f * g * (1.0f + erff(g * 0.70710678118654752f)) * row[j];       // F.gelu (erf form)
    }
    y[(size_t)m * ldy + j] = o;
  }
}

int geglu(float* y, int ldy, const float* in, int M, int F, hipStream_t st) {
  IDX_CHECK(y && in && M > 0 && F > 0 && ldy >= F, "geglu args");
  hipLaunchKernelGGL(geglu_kernel, dim3(M), dim3(256), 0, st, y, ldy, in, F);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void l2norm_scale_kernel(float* y, const float* x, const float* gamma, int d, float scale) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  float ss = 0.0f;
  for (int e = threadIdx.x; e < d; e += 256) { const float v = x[(size_t)m * d + e]; ss = fmaf(v, v, ss); }
  const float nrm = fmaxf(sqrtf(bsum256(ss, red)), 1e-12f);
  for (int e = threadIdx.x; e < d; e += 256) y[(size_t)m * d + e] = x[(size_t)m * d + e] / nrm * scale * gamma[e];
}

int l2norm_scale(float* y, const float* x, const float* gamma, int M, int d, hipStream_t st) {
  IDX_CHECK(y && x && gamma && M > 0 && d > 0, "l2norm args");
  hipLaunchKernelGGL(l2norm_scale_kernel, dim3(M), dim3(256), 0, st, y, x, gamma, d, sqrtf((float)d));
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void lerp_rows_kernel(float* out, const float* base, const float* emo, float alpha, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float bv = base[i]; out[i] = bv + alpha * (emo[i] - bv); }
}

int lerp_rows(float* out, const float* base, const float* emo, float alpha, size_t n, hipStream_t st) {
  IDX_CHECK(out && base && emo, "null pointer");
  if (!n) return 0;
  hipLaunchKernelGGL(lerp_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, out, base, emo, alpha, n);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
