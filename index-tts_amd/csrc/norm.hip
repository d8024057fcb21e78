// Row-wise normalisation with fused residual-stream bookkeeping (one 256-thread workgroup per row).
//
//   x   = x_in[m] (+ add_bias) (+ sum_s partial[s][m])          -> optionally written back (residual stream)
//   y   = norm(x)                                               -> input of the next GEMM / GEMV
// norm modes (reference call sites):
//   NORM_NONE       y = x
//   NORM_LN         torch LayerNorm(eps)                 GPT-2 ln_1 / ln_2 / ln_f (transformers_gpt2.py:615-674, 1171)
//   NORM_LN_LN      LN(g1,b1) then LN(g2,b2)             ln_f followed by final_norm (model_v2.py:208, 562, 611)
//   NORM_ADA_RMS    wmod[b] * (x * rsqrt(mean x^2 + eps) * g1) + bmod[b]
//                                                        AdaptiveLayerNorm(RMSNorm) (gpt_fast/model.py:20-38, 322-333)
//   NORM_MOD_LN     LN(no affine, eps) * (1 + scale[b]) + shift[b]      FinalLayer (diffusion_transformer.py:84-101)
// The split-K partial sum is what lets the skinny decode GEMVs (gemv16.hip) stay atomics-free and
// bitwise reproducible: their K-slices are combined here, in the next kernel's prologue, in a fixed order.
// Reductions: per-thread partial -> wave64 shuffle tree -> one LDS exchange between the 4 waves.
#include "norm.h"
#include "prof.h"

namespace idxtts {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();                 // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

constexpr int NORM_MAX_PER_THREAD = 32;   // d <= 8192

// NPER = elements per thread (compile-time so v[] stays in registers: a runtime-bounded array would
// be demoted to scratch memory)
template <int NPER>
__global__ __launch_bounds__(256) void rows_norm_kernel(const RowsNormArgs p) {
  __shared__ float red[4];
  const int m = blockIdx.x, tid = threadIdx.x, d = p.d;
  constexpr int nper = NPER;
  __bf16* const y_hi = static_cast<__bf16*>(p.y_planes);
  __bf16* const y_lo = y_hi ? y_hi + plane_elems(p.M, d) : nullptr;
  auto put = [&](int e, float val) {      // normalised output: fp32 row and / or split-bf16 planes
    if (p.y) p.y[p.y_frag ? frag_index(m, e, d >> 4) : (size_t)m * p.ld_y + e] = val;
    if (y_hi) {
      const __bf16 hi = (__bf16)val;
      const size_t o = plane_index(m, e, p.M);
      y_hi[o] = hi;
      y_lo[o] = (__bf16)(val - (float)hi);
    }
  };
  float v[NPER];
  const float* xin = nullptr;
  if (p.x_in) {
    if (p.in_rows_per_batch > 0) {
      const int bb = m / p.in_rows_per_batch;
      xin = p.x_in + (size_t)bb * p.in_batch_stride + (size_t)(m - bb * p.in_rows_per_batch) * p.ld_in;
    } else {
      xin = p.x_in + (size_t)m * p.ld_in;
    }
  }
#pragma unroll
  for (int i = 0; i < nper; ++i) {
    const int e = tid + (i << 8);
    float a = 0.0f;
    if (e < d) {
      if (xin) a = p.in_frag ? p.x_in[frag_index(m, e, d >> 4)] : xin[e];
      if (p.add_bias) a += p.add_bias[e];
    }
    v[i] = a;
  }
  if (p.num_partials > 0) {
    // split-K slabs: the decode step has only `rows` workgroups in flight, so memory-level parallelism has to come from
    // inside the thread: groups of 8 slabs x NPER elements are issued back to back before any add (fixed summation order)
    const float* pbase = p.partials + (size_t)m * p.ld_partial;
    const size_t sstride = (size_t)p.partial_rows * p.ld_partial;
    for (int s0 = 0; s0 < p.num_partials; s0 += 8) {
      float t[8][NPER];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < nper; ++i) {
          const int e = tid + (i << 8);
          t[u][i] = (s0 + u < p.num_partials && e < d) ? pbase[(size_t)(s0 + u) * sstride + e] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < nper; ++i) v[i] += t[u][i];
    }
  }
  if (p.x_out) {
#pragma unroll
    for (int i = 0; i < nper; ++i) {
      const int e = tid + (i << 8);
      if (e < d) p.x_out[(size_t)m * p.ld_out + e] = v[i];
    }
  }
  if (p.mode == NORM_NONE || !(p.y || p.y_planes)) {
    if (p.y)
#pragma unroll
      for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) p.y[(size_t)m * p.ld_y + e] = v[i]; }
    return;
  }
  const float inv_d = 1.0f / d;
  const int b = p.rows_per_batch > 0 ? m / p.rows_per_batch : 0;
  if (p.mode == NORM_ADA_RMS) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) ss += v[i] * v[i]; }
    const float r = rsqrtf(block_sum(ss, red) * inv_d + p.eps);
    const float* wm = p.mod_a ? p.mod_a + (size_t)b * p.ld_mod : nullptr;
    const float* bm = p.mod_b ? p.mod_b + (size_t)b * p.ld_mod : nullptr;
#pragma unroll
    for (int i = 0; i < nper; ++i) {
      const int e = tid + (i << 8);
      if (e < d) {
        float o = v[i] * r * p.g1[e];
        if (wm) o = wm[e] * o + bm[e];
        put(e, o);
      }
    }
    return;
  }
  // LayerNorm family: two-pass (mean, then centred variance), as torch does
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) s += v[i]; }
  const float mean = block_sum(s, red) * inv_d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) { const float c = v[i] - mean; ss += c * c; } }
  const float rstd = rsqrtf(block_sum(ss, red) * inv_d + p.eps);
  if (p.mode == NORM_MOD_LN) {
    const float* sh = p.mod_a + (size_t)b * p.ld_mod;   // shift
    const float* sc = p.mod_b + (size_t)b * p.ld_mod;   // scale
#pragma unroll
    for (int i = 0; i < nper; ++i) {
      const int e = tid + (i << 8);
      if (e < d) put(e, (v[i] - mean) * rstd * (1.0f + sc[e]) + sh[e]);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) v[i] = (v[i] - mean) * rstd * p.g1[e] + p.b1[e]; }
  if (p.mode == NORM_LN_LN) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) s += v[i]; }
    const float mean2 = block_sum(s, red) * inv_d;
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) { const float c = v[i] - mean2; ss += c * c; } }
    const float rstd2 = rsqrtf(block_sum(ss, red) * inv_d + p.eps2);
#pragma unroll
    for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) v[i] = (v[i] - mean2) * rstd2 * p.g2[e] + p.b2[e]; }
  }
#pragma unroll
  for (int i = 0; i < nper; ++i) { const int e = tid + (i << 8); if (e < d) put(e, v[i]); }
}

// adaLN-RMSNorm of the DiT (d = 512) straight to split-bf16 planes.  One wave per row (a lane owns 8 consecutive columns: 32-byte
// loads, the row's 2 KiB contiguous; sum of squares by an xor butterfly, no workgroup barrier in the reduction), 16 rows per
// workgroup with all their loads in flight together; the hi / lo images of the 16 rows are staged in LDS chunk-major
// ([32 chunks][16 rows][16 columns] per plane) so that every global store instruction writes whole 512-byte runs of the plane layout
// (plane_index: the 16 rows of a 16-column chunk are adjacent) instead of 32-byte pieces.
// Same arithmetic as the generic kernel's NORM_ADA_RMS branch (sum of squares in a different order).
constexpr int ADA_ROWS = 16;
__global__ __launch_bounds__(256) void ada_rms_planes512_kernel(const RowsNormArgs p) {
  __shared__ __attribute__((aligned(16))) char tile[2 * 32 * ADA_ROWS * 32];      // hi plane image, lo plane image: 16 KiB each
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * ADA_ROWS;
  f32x4 xa[4], xb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wave * 4 + i;
    const float* row = p.x_in + (size_t)min(m, p.M - 1) * p.ld_in + 8 * lane;
    xa[i] = *reinterpret_cast<const f32x4*>(row);
    xb[i] = *reinterpret_cast<const f32x4*>(row + 4);
  }
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.g1 + 8 * lane), gb = *reinterpret_cast<const f32x4*>(p.g1 + 8 * lane + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 4 + i, m = min(m0 + r, p.M - 1);
    const f32x4 a = xa[i], c = xb[i];
    float ss = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rs = rsqrtf(ss * (1.0f / 512.0f) + p.eps);
    f32x4 oa = {a[0] * rs * ga[0], a[1] * rs * ga[1], a[2] * rs * ga[2], a[3] * rs * ga[3]};
    f32x4 ob = {c[0] * rs * gb[0], c[1] * rs * gb[1], c[2] * rs * gb[2], c[3] * rs * gb[3]};
    if (p.mod_a) {
      const int bb = p.rows_per_batch > 0 ? m / p.rows_per_batch : 0;
      const float* wm = p.mod_a + (size_t)bb * p.ld_mod + 8 * lane;
      const float* bm = p.mod_b + (size_t)bb * p.ld_mod + 8 * lane;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wm), w1 = *reinterpret_cast<const f32x4*>(wm + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bm), b1 = *reinterpret_cast<const f32x4*>(bm + 4);
      oa = f32x4{w0[0] * oa[0] + b0[0], w0[1] * oa[1] + b0[1], w0[2] * oa[2] + b0[2], w0[3] * oa[3] + b0[3]};
      ob = f32x4{w1[0] * ob[0] + b1[0], w1[1] * ob[1] + b1[1], w1[2] * ob[2] + b1[2], w1[3] * ob[3] + b1[3]};
    }
    idx_bf16x4 h0, l0, h1, l1;
    split_bf16_x4(oa, h0, l0);
    split_bf16_x4(ob, h1, l1);
    // lane = columns 8 lane .. 8 lane + 7 = half (lane & 1) of chunk (lane >> 1)
    char* dst = tile + (lane >> 1) * (ADA_ROWS * 32) + r * 32 + (lane & 1) * 16;
    *reinterpret_cast<bf16x8_t*>(dst) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8_t*>(dst + 32 * ADA_ROWS * 32) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  __syncthreads();
  char* y_hi = static_cast<char*>(p.y_planes);
  const size_t plane_bytes = plane_elems(p.M, 512) * sizeof(__bf16);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = q * 256 + tid;                 // 16-byte unit: [plane 2][chunk 32][unit 32]
    const int pl = e >> 10, c = (e >> 5) & 31, u = e & 31;
    if (m0 + (u >> 1) < p.M)
      *reinterpret_cast<f32x4*>(y_hi + pl * plane_bytes + ((size_t)c * p.M + m0) * 32 + u * 16) =
          *reinterpret_cast<const f32x4*>(tile + pl * (32 * ADA_ROWS * 32) + c * (ADA_ROWS * 32) + u * 16);
  }
}

int rows_norm_forward(const RowsNormArgs& a, hipStream_t stream) {
  if (a.M == 0) return 0;
  IDX_CHECK(a.M > 0 && a.d > 0 && a.d <= 256 * NORM_MAX_PER_THREAD, "rows_norm shape");
  IDX_CHECK(a.x_in || a.num_partials > 0 || a.add_bias, "rows_norm needs an input");
  if (a.mode == NORM_LN || a.mode == NORM_LN_LN) IDX_CHECK(a.g1 && a.b1, "LayerNorm needs weight and bias");
  if (a.mode == NORM_LN_LN) IDX_CHECK(a.g2 && a.b2, "second LayerNorm needs weight and bias");
  if (a.mode == NORM_ADA_RMS) IDX_CHECK(a.g1 && (!a.mod_a == !a.mod_b), "RMSNorm needs a weight (and both or no modulation vectors)");
  if (a.mode == NORM_MOD_LN) IDX_CHECK(a.mod_a && a.mod_b, "modulated LN needs shift and scale");
  if (a.in_frag || a.y_frag) IDX_CHECK((a.mode == NORM_LN || a.mode == NORM_LN_LN) && a.d % 16 == 0 && a.in_rows_per_batch == 0, "fragment-image I/O: LayerNorm modes, d % 16 == 0");
  const double bytes = 4.0 * a.M * (double)a.d * (1.0 + a.num_partials + (a.x_out ? 1 : 0) + (a.y ? 1 : 0) + (a.y_planes ? 1 : 0));
  static const int cat_ada = prof_register("ada_rms_planes512_kernel"), cat_rows = prof_register("rows_norm_kernel");
  const bool ada512 = a.mode == NORM_ADA_RMS && a.d == 512 && a.y_planes && !a.y && a.x_in && !a.num_partials && !a.add_bias && !a.x_out && !a.in_frag &&
      a.in_rows_per_batch == 0 && (a.ld_in & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.x_in) | reinterpret_cast<uintptr_t>(a.g1) | reinterpret_cast<uintptr_t>(a.mod_a) |
                               reinterpret_cast<uintptr_t>(a.mod_b)) & 15) == 0 && (a.ld_mod & 3) == 0;
  ProfScope prof(ada512 ? cat_ada : cat_rows, stream, 0.0, bytes);
  if (a.mode == NORM_ADA_RMS && a.d == 512 && a.y_planes && !a.y && a.x_in && !a.num_partials && !a.add_bias && !a.x_out && !a.in_frag &&
      a.in_rows_per_batch == 0 && (a.ld_in & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.x_in) | reinterpret_cast<uintptr_t>(a.g1) | reinterpret_cast<uintptr_t>(a.mod_a) |
                               reinterpret_cast<uintptr_t>(a.mod_b)) & 15) == 0 && (a.ld_mod & 3) == 0) {
    hipLaunchKernelGGL(ada_rms_planes512_kernel, dim3(cdiv(a.M, ADA_ROWS)), dim3(256), 0, stream, a);
    IDX_LAUNCH_CHECK();
    return 0;
  }
  const int nper = (a.d + 255) / 256;
  if (nper <= 2) hipLaunchKernelGGL(rows_norm_kernel<2>, dim3(a.M), dim3(256), 0, stream, a);
  else if (nper <= 5) hipLaunchKernelGGL(rows_norm_kernel<5>, dim3(a.M), dim3(256), 0, stream, a);
  else if (nper <= 8) hipLaunchKernelGGL(rows_norm_kernel<8>, dim3(a.M), dim3(256), 0, stream, a);
  else if (nper <= 20) hipLaunchKernelGGL(rows_norm_kernel<20>, dim3(a.M), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(rows_norm_kernel<NORM_MAX_PER_THREAD>, dim3(a.M), dim3(256), 0, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
