// Decode-step skinny GEMM: Y[rows <= 64][N] = epi(LN?(X)[rows][K] . W[K][N]), one HBM pass over the weights, no split-K
// slabs, no LayerNorm launch, no prologue.  Replaces, per generated token, the Conv1D projections of every GPT-2 block
// (transformers_gpt2.py:304-355, 578-592), the LayerNorms in front of c_attn / c_fc (615-674) and the residual adds
// behind c_proj / mlp.c_proj: 5 launches per layer (c_attn, attention, c_proj, c_fc, mlp.c_proj).
//
// What bounds it (tools/stream_probe.hip, tools/gemv_probe.hip on MI355X): a cold 26 MB weight stream alone takes 5.9 us
// (4.5 TB/s) when >= ~256 waves x 5 KiB are in flight; everything else in the kernel is a chain of dependent
// L2 / LDS round trips (~1-2 us each), so the kernel is built to have as few of those as possible:
//   * activations live in HBM as MFMA A-fragment images ([rows/16][K/16][64 lanes][4], frag_index() in common.h): the
//     activation fragment of a 16-k chunk is ONE contiguous 1 KiB load (8 full cache lines), exactly like the weight
//     fragment (packed [N/16][K/16][64][4] for v_mfma_f32_16x16x4_f32), and both are issued together;
//   * a wave owns a K-slice for NTW adjacent column tiles, so one activation fragment feeds NTW weight fragments
//     (L2->L1 activation traffic = weight traffic / NTW);
//   * LayerNorm is folded algebraically: with W' = diag(g) W, u = colsum(W'), c = b.W + bias (precomputed at load),
//         LN(x) . W + bias = rstd * (x . W' - mean * u) + c
//     so the matrix product runs on the RAW residual stream while the row statistics are accumulated from the very
//     fragments the MFMAs consume (shifted sums per wave, Chan's pairwise update across the K-slices in a fixed order:
//     as accurate as the two-pass form, bitwise reproducible);
//   * the K-slices of the 16 waves of a workgroup are reduced through LDS in a fixed order; bias / colsum / residual
//     operands of the epilogue are fetched at kernel entry.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <atomic>

#include "gemv16.h"
#include "prof.h"

namespace idxtts {

// ---- host-side packing: B-fragment order of v_mfma_f32_16x16x4_f32, [N/16][K/16][lane][4], k = 4*(lane>>4)+s, n = lane&15 ----
void pack_gemv16_kn(float* dst, const float* w_kn, int K, int N) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 256;
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
          const int k = c * 16 + 4 * (lane >> 4) + s, n = nt * 16 + (lane & 15);
          sub[lane * 4 + s] = (k < K && n < N) ? w_kn[(size_t)k * N + n] : 0.0f;
        }
    }
}

void pack_gemv16_nk(float* dst, const float* w_nk, int N, int K) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 256;
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
          const int k = c * 16 + 4 * (lane >> 4) + s, n = nt * 16 + (lane & 15);
          sub[lane * 4 + s] = (k < K && n < N) ? w_nk[(size_t)n * K + k] : 0.0f;
        }
    }
}

// ---- compact weight formats (gemv16.h) ----
float fp8_e4m3_decode(unsigned char code) {
  const int e = (code >> 3) & 15, m = code & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);                 // m/8 * 2^-6
  else v = ldexpf(8.0f + (float)m, e - 10);                  // (1 + m/8) * 2^(e-7)
  return (code & 0x80) ? -v : v;
}

unsigned char fp8_e4m3_encode(float v) {
  const unsigned char sign = std::signbit(v) ? 0x80 : 0;
  const float a = std::fabs(v);
  if (!(a < 448.0f)) return sign | 126;                      // saturate (NaN too: quantize_matrix never feeds one)
  if (a < 0.015625f) return sign | (unsigned char)nearbyintf(a * 512.0f);   // subnormal grid 2^-9 (8 -> the smallest normal)
  uint32_t u; memcpy(&u, &a, 4);
  u += 0x7ffffu + ((u >> 20) & 1u);                          // nearest-even on the 3 kept mantissa bits
  const int c = (int)(u >> 20) - ((127 - 7) << 3);
  return sign | (unsigned char)std::min(c, 126);
}

float bf16_round(float v) {
  uint32_t u; memcpy(&u, &v, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return v;
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r; memcpy(&r, &u, 4);
  return r;
}

float fp8_column_scale(float maxabs) {
  if (!(maxabs > 0.0f)) return 1.0f;
  int e; const float f = frexpf(maxabs / 448.0f, &e);        // maxabs/448 = f * 2^e, f in [0.5, 1)
  if (f == 0.5f) e -= 1;
  return ldexpf(1.0f, std::max(-100, std::min(100, e)));
}

void quantize_matrix(float* w, int K, int N, bool kn, int fmt) {
  if (fmt == WFMT_F32) return;
  auto at = [&](int k, int n) -> float& { return kn ? w[(size_t)k * N + n] : w[(size_t)n * K + k]; };
  if (fmt == WFMT_BF16) {
    for (size_t i = 0; i < (size_t)K * N; ++i) w[i] = bf16_round(w[i]);
    return;
  }
  std::vector<float> mx(N, 0.0f);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) mx[n] = std::max(mx[n], std::fabs(at(k, n)));
  for (int n = 0; n < N; ++n) mx[n] = fp8_column_scale(mx[n]);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) { float& v = at(k, n); v = mx[n] * fp8_e4m3_decode(fp8_e4m3_encode(v / mx[n])); }
}

int compact_gemv16(void* dst, const float* pk, int N, int K, int fmt, float* scale_out) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 16);
  const size_t total = (size_t)NT * KC * 256;
  if (fmt == WFMT_BF16) {
    uint16_t* o = static_cast<uint16_t*>(dst);
    for (size_t i = 0; i < total; ++i) {
      uint32_t u; memcpy(&u, &pk[i], 4);
      if (u & 0xffffu) return 1;
      o[i] = (uint16_t)(u >> 16);
    }
    return 0;
  }
  if (fmt != WFMT_FP8) return 1;
  std::vector<float> mx(NT * 16, 0.0f);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      const float* sub = pk + ((size_t)nt * KC + c) * 256;
      for (int i = 0; i < 256; ++i) { float& m = mx[nt * 16 + ((i >> 2) & 15)]; m = std::max(m, std::fabs(sub[i])); }
    }
  for (int n = 0; n < NT * 16; ++n) { mx[n] = fp8_column_scale(mx[n]); if (n < N) scale_out[n] = mx[n]; }
  unsigned char* o = static_cast<unsigned char*>(dst);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      const size_t base = ((size_t)nt * KC + c) * 256;
      for (int i = 0; i < 256; ++i) {
        const float s = mx[nt * 16 + ((i >> 2) & 15)], v = pk[base + i];
        const unsigned char q = fp8_e4m3_encode(v / s);
        if (s * fp8_e4m3_decode(q) != v) return 1;
        o[base + i] = q;
      }
    }
  return 0;
}

__global__ void fp8_decode_table_kernel(float* out) {
  const int c = threadIdx.x;      // 256 codes
  const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(c, false);
  out[c] = lo[0];
}

int fp8_check_device_decode(hipStream_t stream) {
  static bool ok = false;
  if (ok) return 0;
  float* d = nullptr;
  IDX_HIP(hipMalloc(&d, 256 * sizeof(float)));
  hipLaunchKernelGGL(fp8_decode_table_kernel, dim3(1), dim3(256), 0, stream, d);
  float h[256];
  hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d);
  IDX_HIP(e);
  for (int c = 0; c < 256; ++c) {
    const float want = fp8_e4m3_decode((unsigned char)c);
    if (std::isnan(want)) continue;
    if (!(h[c] == want)) IDX_FAIL("this device's fp8 conversion is not OCP e4m3fn (code " + std::to_string(c) + ")");
  }
  ok = true;
  return 0;
}

// Ablation masks of tools/gemv_probe.hip (built with -DGEMV_PROBE); the product build compiles every probe branch out.
#ifdef GEMV_PROBE
#define FX_DBG(mask) (p.dbg & (mask))
#else
#define FX_DBG(mask) false
#endif

struct GemvFXP {
  const float* xf;          // A-fragment images [MT][kc16][64][4]
  const void* wp;           // packed weights [ntiles][kc16][64][4] elements of the WT format
  const float* wscale;      // fp8: [N] power-of-two column scales (null otherwise)
  const float* bias;        // [N] or null (LN-folded layers: c = b.W + bias)
  const float* colsum;      // [N] LN-folded layers: u = colsum(diag(g) W); null = plain GEMV
  float ln_eps;
  const float* res;         // residual, same layout as y, may alias y
  float* y; int y_frag; int ldy;   // y_frag: fragment images with kc16 = N/16 (N % 16 == 0), else row-major [rows][ldy]
  int rows, N, K, kc16, ntiles, kw, cps, act;
  int ksb;                  // > 1: K also split across gridDim.y workgroups; y = raw partial sums [ksb][rows][N] (row-major)
  float* slab;              // fused K-split: partial sums [ksb][rows][N]; the last workgroup of a column tile finishes the job
  unsigned* cnt;            // [gridDim.x] arrival counters (0 on entry and on exit); null = unfused
  int dbg;
};

__device__ __forceinline__ float gelu_new_fx(float v) {
  const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  return 0.5f * v * (1.0f + tanhf(u));
}

// UN = chunks per register batch (two batches live: the next one is in flight while the current one is multiplied)
// SINGLE: the whole K-slice of a wave fits one register batch (every GPT-2 decode shape: K / 16 waves = 5 chunks, the long
// K of mlp.c_proj is split across workgroups): no second register set, which keeps the 16-row kernels at <= 64 VGPRs so
// TWO 1024-thread workgroups fit a CU (8 waves per SIMD) and a 320-tile layer runs in one round.
template <int MT, int NTW, bool SINGLE>
struct FXCfg { static constexpr int UN = SINGLE ? 5 : ((MT + NTW == 2) ? 5 : (NTW == 2 ? 3 : (MT == 2 ? 2 : 1))); };

// What one lane reads of a 16-k weight chunk (4 weights) in each storage format, and how it widens to fp32 (exact)
template <int WT> struct WRaw;
template <> struct WRaw<WFMT_F32> {
  typedef f32x4 raw;
  static __device__ __forceinline__ raw zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4 widen(raw r) { return r; }
};
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <> struct WRaw<WFMT_BF16> {
  typedef u32x2_t raw;
  static __device__ __forceinline__ raw zero() { return raw{0u, 0u}; }
  static __device__ __forceinline__ f32x4 widen(raw r) {
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
};
template <> struct WRaw<WFMT_FP8> {
  typedef uint32_t raw;
  static __device__ __forceinline__ raw zero() { return 0u; }
  static __device__ __forceinline__ f32x4 widen(raw r) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)r, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)r, true);
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
  }
};

// R4 (rows <= 4, MT == 1): the 16-block v_mfma_f32_4x4x1_16B_f32 instead of 16x16x4 -- a quarter of the MFMA cycles for the
// same exact fp32 products.  Block b = lane >> 2 multiplies x[row r][k = 4 (b >> 2) + s] (lane 4 b + r) by
// w[k][n = 4 (b & 3) + c] (lane 4 b + c): the weight stream is read as it is, the activation fragment is read from the lane
// that holds row (lane & 3), and the accumulator ends up as the 16x16 C layout with the four k-groups (lane >> 4) still to add.
// UNS: chunks of the one-batch form.  5 = a K-slice per wave of 16 waves (1024 threads, <= 64 VGPRs: 256 registers per SIMD); 10 = the
// NARROW geometry: 8 waves (512 threads) of 10 chunks each at <= 88 VGPRs -- 176 registers per SIMD, what ONE retiring workgroup of the
// acoustic stage's GEMM / convolution / attention kernels (168 x 1 wave per SIMD) frees on a CU, so a decode launch beside them does
// not have to wait until two of a CU's three workgroups have gone.
template <int MT, int NTW, bool SINGLE, int WT, bool R4, int UNS = 5>
__global__ __launch_bounds__(UNS == 10 ? 512 : 1024) void gemv_fx_kernel(const GemvFXP p) {
  static_assert(!R4 || MT == 1, "the 4-row form has one row tile");
  typedef typename WRaw<WT>::raw wraw_t;
  if (FX_DBG(8)) return;      // tools/gemv_probe.hip: launch + dispatch cost of this geometry alone
  constexpr int UN = SINGLE ? UNS : FXCfg<MT, NTW, SINGLE>::UN;
  constexpr int NB = SINGLE ? 1 : 2;
  constexpr int NACC = MT * NTW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* redbuf = sm;                               // [kw][NACC][256]
  float* wstat = sm + p.kw * NACC * 256;            // [kw][MT*16][2]  per K-slice: mean, M2 (count is known)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nt0 = blockIdx.x * NTW;
  const int c0 = (blockIdx.y * p.kw + wave) * p.cps;
  const int nch = max(0, min(p.cps, p.kc16 - c0));
  const bool ln = p.colsum != nullptr;

  // ---- epilogue operands of this thread's output element, fetched up front ----
  // element e = tid (< NACC*256): r = e & 3, ln16 = (e >> 2) & 63, t = e >> 8 -> (mt, ntw) = (t / NTW, t % NTW)
  const int e_r = tid & 3, e_ln = (tid >> 2) & 63, e_t = tid >> 8;
  const int e_mt = e_t / NTW, e_ntw = e_t - e_mt * NTW;
  const int e_row = e_mt * 16 + (e_ln >> 4) * 4 + e_r, e_col = (nt0 + e_ntw) * 16 + (e_ln & 15);
  const bool e_ok = e_t < NACC && e_row < p.rows && e_col < p.N;
  size_t e_addr = 0;
  float e_bias = 0.f, e_u = 0.f, e_res = 0.f, e_s = 1.f;
  if (e_ok) {
    if (WT == WFMT_FP8) e_s = p.wscale[e_col];
    e_addr = p.y_frag ? frag_index(e_row, e_col, p.N >> 4) : (size_t)e_row * p.ldy + e_col;
    if (p.ksb > 1 && !p.cnt) e_addr = ((size_t)blockIdx.y * p.rows + e_row) * p.N + e_col;
    if (p.bias) e_bias = p.bias[e_col];
    if (ln) e_u = p.colsum[e_col];
    if (p.res) e_res = p.res[e_addr];
  }

  // ---- K loop ----
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Every load of a batch is issued unconditionally, at a clamped chunk index; chunks past this wave's slice are zeroed on the
  // ACTIVATION side when they are consumed.  (A per-element "load or zero" made the compiler branch around each load, and one of
  // the merges became a register copy behind an s_waitcnt vmcnt(0) in the middle of the batch: the last three loads of five were
  // issued only after the first two had come back.)
  const int c0c = min(c0, p.kc16 - 1), last = max(nch - 1, 0);
  const wraw_t* wbase[NTW];      // one raw element = this lane's 4 weights of a chunk; 64 per chunk
#pragma unroll
  for (int j = 0; j < NTW; ++j) wbase[j] = static_cast<const wraw_t*>(p.wp) + ((size_t)min(nt0 + j, p.ntiles - 1) * p.kc16 + c0c) * 64 + lane;
  const float* xbase = p.xf + (size_t)c0c * 256 + (R4 ? ((lane & 48) | (lane & 3)) : lane) * 4;
  const size_t ximg = (size_t)p.kc16 * 256;

  wraw_t wq[NB][UN][NTW];
  f32x4 xq[NB][UN][MT];
  auto load_batch = [&](int buf, int cb) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t idx = (size_t)min(cb + u, last);
#pragma unroll
      for (int j = 0; j < NTW; ++j) wq[buf][u][j] = __builtin_nontemporal_load(wbase[j] + idx * 64);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[buf][u][mt] = *reinterpret_cast<const f32x4*>(xbase + mt * ximg + idx * 256);
    }
  };
  // shifted row sums for the folded LayerNorm: this lane holds row (lane & 15) of every fragment
  float shift[MT], s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) shift[mt] = s1[mt] = s2[mt] = 0.f;

  load_batch(0, 0);
  if (ln) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) shift[mt] = __shfl(xq[0][0][mt][0], lane & 15);   // a sample of the row as the shift
  }
  auto consume = [&](int buf, int cb) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (!(cb + u < nch) || FX_DBG(1)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xq[buf][u][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (ln && cb + u < nch) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float dlt = xq[buf][u][mt][s] - shift[mt];
            s1[mt] += dlt;
            s2[mt] += dlt * dlt;
          }
      }
      f32x4 wf[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) wf[j] = WRaw<WT>::widen(wq[buf][u][j]);
      if (FX_DBG(2)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[mt][j] += xq[buf][u][mt] * wf[j];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
              acc[mt][j] = R4 ? __builtin_amdgcn_mfma_f32_4x4x1f32(xq[buf][u][mt][s], wf[j][s], acc[mt][j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x4f32(xq[buf][u][mt][s], wf[j][s], acc[mt][j], 0, 0, 0);
      }
    }
  };
  if constexpr (SINGLE) {
    consume(0, 0);
  } else {
    for (int cb = 0; cb < nch; cb += 2 * UN) {
      if (cb + UN < nch) load_batch(NB - 1, cb + UN);
      consume(0, cb);
      if (cb + UN < nch) {
        if (cb + 2 * UN < nch) load_batch(0, cb + 2 * UN);
        consume(NB - 1, cb + UN);
      }
    }
  }

  if (FX_DBG(4)) {
    if (acc[0][0][0] == 123.456f) p.y[tid] = acc[0][0][1];
    return;
  }
  if constexpr (R4) {      // add the four k-groups; lanes 0-15 then hold rows 0-3 exactly where the 16x16 layout has them
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[0][j][r];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[0][j][r] = lane < 16 ? v : 0.f;
      }
  }

  // ---- per-slice partial results to LDS (waves beyond kw exist only as epilogue threads) ----
  const bool kwave = wave < p.kw;
  if (kwave)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j) *reinterpret_cast<f32x4*>(&redbuf[((wave * NACC + mt * NTW + j) * 64 + lane) * 4]) = acc[mt][j];
  if (ln && kwave && !FX_DBG(16)) {
    // the four lane groups (lane >> 4) hold disjoint k of the same row: fold them, then slice mean / M2 about the mean
    const float cnt = 16.0f * nch;    // K-slice elements per row (padded chunks excluded: K % 16 == 0 for folded layers)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float a = s1[mt], b = s2[mt];
      a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
      a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
      if (lane < 16) {
        const float mean_w = nch > 0 ? shift[mt] + a / cnt : 0.f;
        const float m2_w = nch > 0 ? b - a * a / cnt : 0.f;
        wstat[(wave * MT * 16 + mt * 16 + lane) * 2 + 0] = mean_w;
        wstat[(wave * MT * 16 + mt * 16 + lane) * 2 + 1] = m2_w;
      }
    }
  }
  __syncthreads();
  // Row statistics of the folded LayerNorm, combined per WAVE for the 4 rows its 64 epilogue threads own (no serial pass, no
  // second barrier): lane j holds K-slice (j >> 2) of its own row (row = ... + (j & 3), see e_row), an xor butterfly over the
  // slice bits adds them -- the same tree in every lane, so all lanes of a row agree bit for bit -- with the exact identity
  //   mean = sum_w n_w mean_w / n,   M2 = sum_w M2_w + sum_w n_w (mean_w - mean)^2
  float ln_mean = 0.f, ln_rstd = 0.f;
  if (ln && e_t < NACC && !FX_DBG(32)) {      // wave-uniform (e_t = tid >> 8); rows beyond p.rows are zero padding: harmless
    const int sl = lane >> 2;
    const bool on = sl < p.kw;
    const float nw = on ? 16.0f * max(0, min(p.cps, p.kc16 - sl * p.cps)) : 0.f;      // (folded LayerNorm implies ksb == 1)
    const int srow = e_mt * 16 + (e_ln >> 4) * 4 + e_r;
    const float2 st = on ? *reinterpret_cast<const float2*>(&wstat[(sl * MT * 16 + srow) * 2]) : make_float2(0.f, 0.f);
    float n = nw, a = nw * st.x;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { n += __shfl_xor(n, m); a += __shfl_xor(a, m); }
    ln_mean = a / n;
    const float dlt = st.x - ln_mean;
    float m2 = st.y + nw * dlt * dlt;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) m2 += __shfl_xor(m2, m);
    ln_rstd = rsqrtf(m2 / n + p.ln_eps);
  }

  // ---- fixed-order reduction over the K-slices and the epilogue: one output element per thread ----
  if (p.ksb > 1 && p.cnt) {
    // K split across workgroups, finished by whichever of them arrives last (wait-free: nobody spins).  Partial sums travel
    // through agent-scope atomics (coherent across the XCDs' L2s); the sum runs in slab order, so the result does not depend
    // on who is last.
    __shared__ int s_last;
    if (e_ok) {
      float v = 0.f;
      for (int w = 0; w < p.kw; ++w) v += redbuf[(w * NACC + e_t) * 256 + (tid & 255)];
      if (WT == WFMT_FP8) v *= e_s;
      __hip_atomic_store(&p.slab[((size_t)blockIdx.y * p.rows + e_row) * p.N + e_col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // EVERY wave waits for the acknowledgement of its device-scope (sc1) stores before the workgroup barrier, so the arrival
    // below is issued after all partial sums of this workgroup are at the device coherence point.  (An acq_rel arrival would
    // say the same in the memory model, but costs an L2 write-back + invalidate per launch: measured +10 us.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&p.cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)p.ksb - 1u;
      if (s_last) __hip_atomic_store(&p.cnt[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last || !e_ok) return;
    float t[8];      // all partial sums in ONE round trip (a loop of atomic loads is issued one after the other)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      t[s] = s < p.ksb ? __hip_atomic_load(&p.slab[((size_t)s * p.rows + e_row) * p.N + e_col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    float v = p.res ? e_res : 0.f;
    v += e_bias;
#pragma unroll
    for (int s = 0; s < 8; ++s) v += t[s];
    p.y[e_addr] = v;
    return;
  }
  if (e_t < NACC) {
    float v = 0.f;
    for (int w = 0; w < p.kw; ++w) v += redbuf[(w * NACC + e_t) * 256 + (tid & 255)];
    if (e_ok) {
      if (WT == WFMT_FP8) v *= e_s;        // power-of-two column scale: exact
      if (ln) v = ln_rstd * (v - ln_mean * e_u);
      v += e_bias;
      if (p.act == 1) v = gelu_new_fx(v);
      if (p.res) v += e_res;
      p.y[e_addr] = v;
    }
  }
}

// Narrow layers (N = d: 80 column tiles) with a long K (mlp.c_proj: 26 MB) cannot reach HBM speed from 80 workgroups:
// a CU sustains ~11 B/clk from HBM (tools/l2_probe.hip), so all 256 have to stream.  Their K is split across `ksb`
// workgroups per column tile; the partial sums go to a [ksb][rows][N] slab and gemv_fx_combine adds them in a fixed order
// (bitwise reproducible) together with bias and residual.
int gemv_fx_ksb(int N, int K) {
  const int ntiles = cdiv(N, 16), kc16 = cdiv(K, 16);
  return (ntiles <= 128 && kc16 >= 64) ? 4 : 1;
}

__global__ __launch_bounds__(256) void gemv_fx_combine_kernel(const float* __restrict__ slab, int ksb, int rows, int N, const float* __restrict__ bias,
                                                              const float* res, float* y) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * N) return;
  const int row = idx / N, col = idx - row * N;
  float t[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) t[s] = s < ksb ? slab[((size_t)s * rows + row) * N + col] : 0.f;
  const size_t o = frag_index(row, col, N >> 4);
  float v = res ? res[o] : 0.f;
  if (bias) v += bias[col];
#pragma unroll
  for (int s = 0; s < 8; ++s) v += t[s];
  y[o] = v;
}

// y (fragment images over N) = res + bias + sum_s slab[s]
int gemv_fx_combine(const float* slab, int ksb, int rows, int N, const float* bias, const float* res, float* y, hipStream_t stream) {
  IDX_CHECK(slab && y && ksb >= 1 && ksb <= 8 && N % 16 == 0, "combine arguments");
  static const int cat = prof_register("gemv_fx_combine_kernel");
  ProfScope prof(cat, stream, 0.0, 4.0 * rows * (double)N * (ksb + 2));
  hipLaunchKernelGGL(gemv_fx_combine_kernel, dim3(cdiv(rows * N, 256)), dim3(256), 0, stream, slab, ksb, rows, N, bias, res, y);
  IDX_LAUNCH_CHECK();
  return 0;
}

// Process-wide geometry of the 5..16-row decode GEMVs on compact weight streams (include/idxtts.h::idxtts_set_decode_geometry).
static std::atomic<int> g_decode_narrow{0};
void set_decode_geometry(int narrow) { g_decode_narrow.store(narrow ? 1 : 0); }
int get_decode_geometry() { return g_decode_narrow.load(); }
static bool gemv_fx_narrow() { return g_decode_narrow.load() != 0; }

static void gemv_fx_plan_geom(int N, int K, int rows, bool narrow_geom, int* ntw, int* kw) {
  const int kc16 = cdiv(K, 16), ntiles = cdiv(N, 16), MT = cdiv(rows, 16);
  int k = 16;
  while (k > 1 && kc16 < k) k >>= 1;
  *kw = k;
  if (narrow_geom && MT == 1 && rows > 4 && kc16 % 80 == 0) { *kw = 8; *ntw = 1; return; }
  // Two column tiles per wave (one activation fragment feeds both) exactly when that turns a two-round launch into one
  // round of <= 256 workgroups (c_fc: 320 tiles); measured per shape in profiles/r01_gemv_probe.txt
  *ntw = (MT <= 2 && ntiles > 256 && cdiv(ntiles, 2) <= 256) ? 2 : 1;
}

void gemv_fx_plan(int N, int K, int rows, int* ntw, int* kw) { gemv_fx_plan_geom(N, K, rows, gemv_fx_narrow(), ntw, kw); }

int gemv_fx_forward(const Gemv16Weights& w, const GemvFXArgs& a, hipStream_t stream) {
  const bool narrow_geom = gemv_fx_narrow();      // the process-wide switch, read ONCE per launch (a setter racing with a launch cannot tear the plan)
  IDX_CHECK(w.wp && a.xf && a.y, "null pointer");
  IDX_CHECK(a.rows > 0 && a.rows <= 64, "1..64 rows");
  IDX_CHECK((reinterpret_cast<uintptr_t>(a.xf) & 15) == 0, "x alignment");
  if (a.colsum) IDX_CHECK(w.K % 16 == 0, "folded LayerNorm needs K % 16 == 0");
  if (a.y_frag) IDX_CHECK(w.N % 16 == 0, "fragment-image output needs N % 16 == 0");
  GemvFXP p;
  IDX_CHECK(w.fmt == WFMT_F32 || w.fmt == WFMT_BF16 || (w.fmt == WFMT_FP8 && w.wscale), "weight format");
  p.xf = a.xf; p.wp = w.wp; p.wscale = w.wscale; p.bias = a.bias; p.colsum = a.colsum; p.ln_eps = a.ln_eps;
  p.res = a.res; p.y = a.y; p.y_frag = a.y_frag; p.ldy = a.ldy;
  p.rows = a.rows; p.N = w.N; p.K = w.K; p.kc16 = cdiv(w.K, 16); p.ntiles = cdiv(w.N, 16);
  int ntw = 1;
  gemv_fx_plan_geom(w.N, w.K, a.rows, narrow_geom, &ntw, &p.kw);
  p.ksb = a.ksb > 1 ? a.ksb : 1;
  p.slab = a.slab; p.cnt = p.ksb > 1 ? a.ksb_counters : nullptr;
  if (p.ksb > 1 && !p.cnt) IDX_CHECK(!a.colsum && !a.bias && !a.res && a.act == 0 && !a.y_frag, "a K-split launch writes raw partial sums");
  if (p.cnt) IDX_CHECK(a.slab && !a.colsum && a.act == 0 && p.ksb <= 8, "fused K-split: slab, no folded LayerNorm, no activation, <= 8 pieces");
  p.cps = cdiv(p.kc16, p.kw * p.ksb);
  p.act = a.act; p.dbg = a.dbg;
  const int MT = cdiv(a.rows, 16);
  const bool narrow = p.kw == 8 && p.cps == 10 && MT == 1 && a.rows > 4 && w.fmt != WFMT_F32 && narrow_geom;
  if (p.kw == 8 && !narrow && narrow_geom && MT == 1 && a.rows > 4 && cdiv(w.K, 16) % 80 == 0) {      // fp32 streams: the 16-wave form
    p.kw = 16; p.cps = cdiv(p.kc16, p.kw * p.ksb);
    ntw = (p.ntiles > 256 && cdiv(p.ntiles, 2) <= 256) ? 2 : 1;
  }
  const bool single = p.cps <= 5;
  if (MT == 2 && !single) ntw = 1;      // two column tiles per wave at 17..32 rows exist in the one-batch form only
  const int nacc = MT * ntw;
  const int threads = std::max(64 * p.kw, 256 * nacc);      // one epilogue thread per output element of the workgroup
  IDX_CHECK(threads <= 1024, "workgroup size");
  const size_t lds = (size_t)(p.kw * nacc * 256 + p.kw * MT * 32 + MT * 32) * sizeof(float);
  dim3 grid(cdiv(p.ntiles, ntw), p.ksb);
  const double flops = 2.0 * a.rows * (double)w.N * w.K;
  const double bytes = (double)wfmt_bytes(w.fmt) * w.N * w.K + 4.0 * ((double)a.rows * w.N * (a.res ? 2.0 : 1.0) + (double)a.rows * w.K);
  static const int cat = prof_register("gemv_fx_kernel");
  ProfScope prof(cat, stream, flops, bytes);
#define LAUNCH_R(MTV, NTWV, SG, WTV, R4V)                                                                                 \
  {                                                                                                                       \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_fx_kernel<MTV, NTWV, SG, WTV, R4V>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));                               \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemv_fx_kernel<MTV, NTWV, SG, WTV, R4V>), grid, dim3(threads), lds, stream, p);                   \
  }
#define LAUNCH_N(WTV)                                                                                                     \
  {                                                                                                                       \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_fx_kernel<1, 1, true, WTV, false, 10>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));                               \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemv_fx_kernel<1, 1, true, WTV, false, 10>), grid, dim3(threads), lds, stream, p);                \
  }
#define LAUNCH_W(MTV, NTWV, SG, WTV)                                                                                      \
  {                                                                                                                       \
    if (MTV == 1 && r4) LAUNCH_R(1, NTWV, SG, WTV, true)                                                                  \
    else LAUNCH_R(MTV, NTWV, SG, WTV, false)                                                                              \
  }
#define LAUNCH(MTV, NTWV, SG)                                                                                             \
  {                                                                                                                       \
    if (w.fmt == WFMT_FP8) LAUNCH_W(MTV, NTWV, SG, WFMT_FP8)                                                              \
    else if (w.fmt == WFMT_BF16) LAUNCH_W(MTV, NTWV, SG, WFMT_BF16)                                                       \
    else LAUNCH_W(MTV, NTWV, SG, WFMT_F32)                                                                                \
  }
  const bool r4 = a.rows <= 4;
  if (narrow) {
    if (w.fmt == WFMT_FP8) { LAUNCH_N(WFMT_FP8) } else { LAUNCH_N(WFMT_BF16) }
  } else
  if (MT == 1 && ntw == 2 && single) LAUNCH(1, 2, true)
  else if (MT == 1 && ntw == 2) LAUNCH(1, 2, false)
  else if (MT == 1 && single) LAUNCH(1, 1, true)
  else if (MT == 1) LAUNCH(1, 1, false)
  // 17..48 rows (coalesced decodes, 16 utterances x 3 beams): the one-batch form as well -- all five chunks of a wave's K-slice in
  // flight at once (10 + 20 MT VGPRs of operands: one workgroup per CU); the two-buffer form with 1-2 chunks per batch measured
  // 14.6 / 16.5 us per launch at 32 / 48 rows against 7.8 at 16 (profiles/README.md "Round 3")
  else if (MT == 2 && ntw == 2 && single) LAUNCH(2, 2, true)
  else if (MT == 2 && single) LAUNCH(2, 1, true)
  else if (MT == 3 && single) LAUNCH(3, 1, true)
  else if (MT == 2) LAUNCH(2, 1, false) else if (MT == 3) LAUNCH(3, 1, false) else LAUNCH(4, 1, false)
#undef LAUNCH_N
#undef LAUNCH_R
#undef LAUNCH_W
#undef LAUNCH
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
