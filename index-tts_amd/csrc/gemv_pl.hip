// Decode-step skinny GEMM on the bf16 matrix pipe (gemv_pl.h): replaces, per generated token and for up to 64 rows at once, the Conv1D
// projections of every GPT-2 block (transformers_gpt2.py:304-355, 578-592), the LayerNorms in front of c_attn / c_fc (615-674), the
// residual adds behind c_proj / mlp.c_proj and mel_head -- the reference's accel engine batches its decode rows into one graph replay
// the same way (accel/accel_engine.py:221-310, 358-376).
//
// What decides the shape of the kernel at 48-64 rows (MI355X): a 16-column weight fragment (1 KiB per 32 k) is multiplied by 3 * MT
// activation fragments (3 planes x MT row tiles, 1 KiB each) -- the ACTIVATION side is 9-12 x the bytes of the weight side per
// workgroup, a CU takes ~70 GB/s from its L2 and ~24 GB/s from HBM, and all 256 CUs have to stream weights.  So
//   * a workgroup covers CT column tiles x a K part of KW * 5 chunks; the K part of the activations goes L2 -> LDS ONCE per workgroup
//     (LDS-DMA, one 1-KiB fragment per wave-instruction) and is read from there by all eight waves;
//   * K is split across the waves of a workgroup (KW) AND across workgroups (grid.y parts), so a launch still has 240-320 workgroups;
//     the parts' partial sums (MFMA C layout, whole 1-KiB write-through stores) are added in part order by the last wave to arrive at
//     the column tile's counter -- wait-free and bitwise reproducible whoever is last;
//   * every weight load of a wave is issued at kernel entry (non-temporal: read once), before the activation DMA;
//   * the folded LayerNorm's row statistics are not recomputed by 240 workgroups from the activations: the producer of x leaves
//     per-16-column partials (mean, M2), combined here with Chan's update while the weight stream is in flight;
//   * outputs leave as fp32 rows and / or as the three bf16 planes of the next GEMV (16-byte runs through a wave-private LDS transpose).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemv_pl.h"
#include "prof.h"

namespace idxtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

int pack_gemv32(void* dst, const float* w, int N, int K, bool kn, int fmt, float* scale_out) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 32);
  auto at = [&](int k, int n) -> float { return (k < K && n < N) ? (kn ? w[(size_t)k * N + n] : w[(size_t)n * K + k]) : 0.0f; };
  if (fmt == WFMT_BF16) {
    uint16_t* o = static_cast<uint16_t*>(dst);
    for (int nt = 0; nt < NT; ++nt)
      for (int c = 0; c < KC; ++c)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const float v = at(c * 32 + 8 * (lane >> 4) + j, nt * 16 + (lane & 15));
            uint32_t u; memcpy(&u, &v, 4);
            if (u & 0xffffu) return 1;
            o[(((size_t)nt * KC + c) * 64 + lane) * 8 + j] = (uint16_t)(u >> 16);
          }
    return 0;
  }
  if (fmt != WFMT_FP8) return 1;
  std::vector<float> sc(NT * 16, 1.0f);
  for (int n = 0; n < N; ++n) {
    float mx = 0.0f;
    for (int k = 0; k < K; ++k) mx = std::max(mx, std::fabs(at(k, n)));
    sc[n] = fp8_column_scale(mx);
    if (scale_out) scale_out[n] = sc[n];
  }
  unsigned char* o = static_cast<unsigned char*>(dst);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int n = nt * 16 + (lane & 15);
          const float v = at(c * 32 + 8 * (lane >> 4) + j, n);
          const unsigned char q = fp8_e4m3_encode(v / sc[n]);
          if (sc[n] * fp8_e4m3_decode(q) != v) return 1;
          o[(((size_t)nt * KC + c) * 64 + lane) * 8 + j] = q;
        }
  return 0;
}

struct GemvPLP {
  const float* x; int ldx;       // [rows][ldx] fp32
  const void* wp;                // [ntiles][KC][64][8] bf16 | fp8
  const float* wscale;
  const float* bias; const float* colsum; float ln_eps;
  const float* stats_in; int stats_tiles;
  const float* res; float* y; int ldy;
  float* stats_out;
  float* slab; unsigned* cnt;
  int rows, N, K, KC, ntiles, act, kparts;
  int slab_bytes;
#ifdef PL_STAMPS
  unsigned long long* stamps;      // diagnostic build of tools/gemv_pl_probe.hip only: [workgroup][16] s_memtime / s_memrealtime stamps
#endif
};
#ifdef PL_STAMPS
static unsigned long long* g_pl_stamps = nullptr;
void gemv_pl_set_stamps(unsigned long long* buf) { g_pl_stamps = buf; }
#define PLS(k) do { if (p.stamps && tid == 0) p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PLR(k) do { if (p.stamps && tid == 0) p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PLS(k)
#define PLR(k)
#endif

// gelu_new with tanh(u) = 1 - 2 / (e^{2u} + 1) on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7 in tanh: inside fp32 rounding of the product)
__device__ __forceinline__ float gelu_new_pl(float v) {
  const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);
  const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  return 0.5f * v * (1.0f + th);
}

// all-reduce (+) over the 16 lanes of a DPP row by rotations: four VALU adds, no LDS crossbar round trip (ds_bpermute)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));      // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));      // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));      // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));      // row_ror:1
  return v;
}
// (n, mean, M2) of a set and of another, disjoint one -> of their union (Chan et al.); empty sets allowed; branch-free, the one
// reciprocal by v_rcp_f32 (deterministic: every workgroup computes the same function of the same partials in the same order)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, const float nb, const float mb, const float m2b) {
  const float nn = n + nb;
  const float f = nb * __builtin_amdgcn_rcpf(fmaxf(nn, 1.0f));
  const float dlt = mb - mean;
  mean += dlt * f;
  m2 += m2b + dlt * dlt * n * f;
  n = nn;
}

template <int WT> struct PLW;
template <> struct PLW<WFMT_BF16> {
  typedef u32x4 raw;
  static __device__ __forceinline__ bf16x8 widen(raw r) { return __builtin_bit_cast(bf16x8, r); }
};
template <> struct PLW<WFMT_FP8> {
  typedef u32x2 raw;
  // e4m3 -> fp32 is exact and every e4m3 value is a bf16 value: keep the upper halves
  static __device__ __forceinline__ bf16x8 widen(raw r) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], true);
      o[2 * i] = __builtin_amdgcn_perm(__float_as_uint(lo[1]), __float_as_uint(lo[0]), 0x07060302u);
      o[2 * i + 1] = __builtin_amdgcn_perm(__float_as_uint(hi[1]), __float_as_uint(hi[0]), 0x07060302u);
    }
    return __builtin_bit_cast(bf16x8, o);
  }
};

constexpr int PL_CPS = 5;      // 32-k chunks per wave: 8 waves x 5 chunks x 32 = the GPT's d = 1280 in one workgroup, 4 d in four

// LDS: [A image: K part x 3 planes x MT fragments of 1 KiB][row statistics 8 KiB][K-slice partial sums + finished tiles: 8 MT KiB]
// (the last region only where the kernel uses it: K split over the waves of a workgroup, or no K split across workgroups)
static constexpr int pl_lds_bytes(int MT, int CT, bool with_out) {
  const int KW = 8 / CT, kpart = KW * PL_CPS;
  return kpart * 3 * MT * 1024 + 8192 + (with_out ? 8 * MT * 1024 : 0);
}

// four fp32 -> the three bf16 planes (h = top 8 significant bits, m the next 8, l the last 8; exact), each four bf16 = 8 bytes
__device__ __forceinline__ void split3_x4(const f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
  unsigned hu[4], mu[4], lu[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hu[e] = __float_as_uint(v[e]);
    const float r1 = v[e] - __uint_as_float(hu[e] & 0xffff0000u);
    mu[e] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(mu[e] & 0xffff0000u);
    lu[e] = __float_as_uint(r2);
  }
  h = u32x2{__builtin_amdgcn_perm(hu[1], hu[0], 0x07060302u), __builtin_amdgcn_perm(hu[3], hu[2], 0x07060302u)};
  m = u32x2{__builtin_amdgcn_perm(mu[1], mu[0], 0x07060302u), __builtin_amdgcn_perm(mu[3], mu[2], 0x07060302u)};
  l = u32x2{__builtin_amdgcn_perm(lu[1], lu[0], 0x07060302u), __builtin_amdgcn_perm(lu[3], lu[2], 0x07060302u)};
}

template <int MT, int CT, int WT>
__global__ __launch_bounds__(512) void gemv_pl_kernel(const GemvPLP p) {
  typedef typename PLW<WT>::raw wraw_t;
  constexpr int KW = 8 / CT, CPS = PL_CPS, KPART = KW * CPS, R = MT * 16;
  constexpr int NFRAG = KPART * 3 * MT;
  constexpr int QPR = KPART * 8;                    // 4-element groups of a row inside the K part
  constexpr int NQJ = (QPR + 31) / 32;              // ... per thread and row tile
  constexpr int NU = CT * MT, UPW = (NU + 7) / 8;   // (column tile, row tile) units of the epilogue, per wave
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* const aimg = sm;                                                      // [KPART][3 planes][MT] fragments of 1 KiB
  float* const stat = reinterpret_cast<float*>(sm + NFRAG * 1024);            // [64][2] mean, rstd per row; [8 waves][64 rows][3] partials behind
  float* const stp = stat + 128;
  float* const red = stat + 2048;                                            // [KW - 1][CT][MT][64][4] K-slice partial sums, then
  float* const outi = red + (KW - 1) * CT * MT * 256;                         // [CT][MT][64][4] finished tiles (no K split across workgroups)
  int& s_last = *reinterpret_cast<int*>(stat + 2040);      // (inside the statistics block: no static LDS beside the dynamic image)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ct = wave % CT, kw = wave / CT;
  const int tile = blockIdx.x * CT + ct;
  const int tile_c = min(tile, p.ntiles - 1);
  const int kp = blockIdx.y;
  const int cbase = kp * KPART, c0 = cbase + kw * CPS;
  const bool ln = p.colsum != nullptr;
  const int lastc = p.KC - 1;

  PLR(0); PLS(1);
  // ---- 2. the activations of this workgroup's K part: fp32 rows from L2.  A wave-instruction covers 16 rows x 64 bytes; thread ->
  //         row (tid & 15) of every row tile, 4-element groups (tid >> 4) + 32 j of the K part ----
  f32x4 xq[MT][NQJ];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NQJ; ++j) {
      const int q = (tid >> 4) + 32 * j, row = mt * 16 + (tid & 15), k = cbase * 32 + 4 * q;
      const bool on = q < QPR && row < p.rows && k < p.K;
      xq[mt][j] = *reinterpret_cast<const f32x4*>(p.x + (size_t)(on ? row : 0) * p.ldx + (on ? k : 0));
    }

  // ---- 3. row statistics of the folded LayerNorm: the producer's per-16-column partials [tile][row] (mean, M2); lane = row (one
  //         512-byte run per wave-instruction), wave w takes tiles w, w + 8, ... ----
  float2 st0[10];
  const float2* const sp = reinterpret_cast<const float2*>(p.stats_in) + min(lane, R - 1);
  if (ln) {
#pragma unroll
    for (int i = 0; i < 10; ++i) st0[i] = sp[(size_t)min(wave + 8 * i, p.stats_tiles - 1) * R];
  }

  // ---- 4. epilogue operands of this wave's (column tile, row tile) units: unit u = wave + 8 i -> tile u / MT, row tile u % MT.
  //         Every load unconditional at a clamped address (a load under its own branch makes the compiler drain vmcnt at the join) ----
  const int g4 = (lane >> 4) * 4;
  float e_bias[UPW], e_u[UPW], e_s[UPW], e_res[UPW][4];
  {
    const float* const bp = p.bias ? p.bias : p.x;
    const float* const up = ln ? p.colsum : p.x;
    const float* const sp8 = WT == WFMT_FP8 ? p.wscale : p.x;
    const float* const rp = p.res ? p.res : p.y;
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int u = min(wave + 8 * i, NU - 1), uct = u / MT, umt = u - uct * MT;
      const int col = min((blockIdx.x * CT + uct) * 16 + (lane & 15), p.N - 1);
      e_bias[i] = bp[p.bias ? col : 0];
      e_u[i] = up[ln ? col : 0];
      e_s[i] = sp8[WT == WFMT_FP8 ? col : 0];
#pragma unroll
      for (int r = 0; r < 4; ++r) e_res[i][r] = rp[(size_t)min(umt * 16 + g4 + r, p.rows - 1) * p.ldy + col];
    }
  }

  // ---- 1. this wave's weight fragments (read once: non-temporal) -- the long pole, issued last so that everything the LDS image
  //         needs (L2 hits) can be consumed while they are still in flight (vmcnt retires in order) ----
  wraw_t wq[CPS];
  {
    const wraw_t* wb = static_cast<const wraw_t*>(p.wp) + (size_t)tile_c * p.KC * 64 + lane;
#pragma unroll
    for (int u = 0; u < CPS; ++u) wq[u] = __builtin_nontemporal_load(wb + (size_t)min(c0 + u, lastc) * 64);
  }
  __builtin_amdgcn_sched_barrier(0);      // every load above is issued before anything below waits for one
  PLS(2);
  // ---- 5. activations -> three bf16 planes in A-fragment order (lane = row + 16 (k % 32) / 8, 8 k per lane) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NQJ; ++j) {
      const int q = (tid >> 4) + 32 * j, row = mt * 16 + (tid & 15), k = cbase * 32 + 4 * q;
      if (q < QPR) {
        const bool on = row < p.rows && k < p.K;
        u32x2 h, m, l;
        split3_x4(on ? xq[mt][j] : f32x4{0.f, 0.f, 0.f, 0.f}, h, m, l);
        char* dst = aimg + (size_t)(((q >> 3) * 3) * MT + mt) * 1024 + ((tid & 15) + 16 * ((q & 7) >> 1)) * 16 + (q & 1) * 8;
        *reinterpret_cast<u32x2*>(dst) = h;
        *reinterpret_cast<u32x2*>(dst + MT * 1024) = m;
        *reinterpret_cast<u32x2*>(dst + 2 * MT * 1024) = l;
      }
    }
  if (ln) {
    // this wave's tiles (16 columns each) for row `lane`: mean of the means, M2 = sum M2_j + 16 sum (mean_j - mean)^2 -- no division
    // chain; the 8 waves' results are combined per row behind the barrier (fixed order)
    float sn = 0.f, smean = 0.f, sm2 = 0.f;
    for (int j0 = 0; j0 < p.stats_tiles; j0 += 80) {
      float2 t[10];
      if (j0 == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) t[i] = st0[i];
      } else {
#pragma unroll
        for (int i = 0; i < 10; ++i) t[i] = sp[(size_t)min(j0 + wave + 8 * i, p.stats_tiles - 1) * R];
      }
      float cnt = 0.f, S = 0.f, Q = 0.f;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const bool on = j0 + wave + 8 * i < p.stats_tiles;
        cnt += on ? 1.f : 0.f; S += on ? t[i].x : 0.f; Q += on ? t[i].y : 0.f;
      }
      const float mloc = S * __builtin_amdgcn_rcpf(fmaxf(cnt, 1.f));
      float D = 0.f;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const float dl = (j0 + wave + 8 * i < p.stats_tiles) ? t[i].x - mloc : 0.f;
        D += dl * dl;
      }
      chan_merge(sn, smean, sm2, 16.f * cnt, mloc, Q + 16.f * D);
    }
    stp[(wave * 64 + lane) * 3 + 0] = sn; stp[(wave * 64 + lane) * 3 + 1] = smean; stp[(wave * 64 + lane) * 3 + 2] = sm2;
  }
  PLS(3);
  __syncthreads();
  PLS(4);
  if (ln && wave == 7) {      // rows' statistics: the 8 waves' partials in wave order (read by the epilogue, behind further barriers)
    float n8[8], m8[8], q8[8];      // a fixed tree ((0,1),(2,3)),((4,5),(6,7)): three dependent merges instead of eight
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) { n8[w2] = stp[(w2 * 64 + lane) * 3]; m8[w2] = stp[(w2 * 64 + lane) * 3 + 1]; q8[w2] = stp[(w2 * 64 + lane) * 3 + 2]; }
#pragma unroll
    for (int st = 1; st < 8; st <<= 1)
#pragma unroll
      for (int w2 = 0; w2 < 8; w2 += 2 * st) chan_merge(n8[w2], m8[w2], q8[w2], n8[w2 + st], m8[w2 + st], q8[w2 + st]);
    stat[lane * 2 + 0] = m8[0];
    stat[lane * 2 + 1] = rsqrtf(q8[0] * __builtin_amdgcn_rcpf(fmaxf(n8[0], 1.f)) + p.ln_eps);
  }

  // ---- 6. products: 3 planes x MT row tiles per weight fragment ----
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CPS; ++u) {
    if (c0 + u <= lastc) {
      const bf16x8 b = PLW<WT>::widen(wq[u]);
      const char* fr = aimg + (size_t)((kw * CPS + u) * 3 * MT) * 1024 + lane * 16;
#pragma unroll
      for (int pl = 2; pl >= 0; --pl)        // low-order plane first
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(fr + (pl * MT + mt) * 1024), b, acc[mt], 0, 0, 0);
    }
  }

  PLS(5);
  // ---- 7. the K slices of this workgroup's waves ----
  if constexpr (KW > 1) {
    if (kw > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(&red[((((kw - 1) * CT + ct) * MT + mt) * 64 + lane) * 4]) = acc[mt];
    }
    __syncthreads();
    if (kw == 0) {
#pragma unroll
      for (int k2 = 1; k2 < KW; ++k2)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] += *reinterpret_cast<const f32x4*>(&red[((((k2 - 1) * CT + ct) * MT + mt) * 64 + lane) * 4]);
    }
  }

  // ---- 8. the K parts of other workgroups: partial sums in C layout (whole 1-KiB write-through stores); the LAST workgroup of
  //         this column group to arrive adds them in part order -- wait-free, bitwise reproducible whoever is last ----
  PLS(6);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, p.slab_bytes, 0x00020000);
  const int part_stride = p.ntiles * MT * 1024;      // bytes
  if (p.kparts > 1) {
    if (kw == 0 && tile < p.ntiles) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt]), rs, (tile * MT + mt) * 1024 + lane * 16, kp * part_stride, 17);      // sc0 sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // acknowledged at the device coherence point before the arrival
    PLS(7);
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&p.cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)p.kparts - 1u;
      if (s_last) __hip_atomic_store(&p.cnt[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    PLS(8); PLR(9);
    if (!s_last) return;
  } else {
    if (kw == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(&outi[((ct * MT + mt) * 64 + lane) * 4]) = acc[mt];
    }
    __syncthreads();
  }

  // ---- 9. epilogue, one (column tile, row tile) unit at a time, the units spread over the 8 waves; C layout: lane = column
  //         (lane & 15), rows 4 (lane >> 4) + r ----
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i, uct = u / MT, umt = u - uct * MT;
    const int utile = blockIdx.x * CT + uct;
    if (u >= NU || utile >= p.ntiles) continue;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (p.kparts > 1) {
      const int mine = (utile * MT + umt) * 1024 + lane * 16;
      // Summation order over K, the same for every geometry: groups of 5 chunks (one wave's slice), neighbouring groups added
      // pairwise, the pairs accumulated left to right -- with two slices per workgroup a part IS a pair, with one slice per workgroup
      // two parts make one; so a row's result does not depend on the geometry the row count selects
      for (int s0 = 0; s0 < p.kparts; s0 += 8) {
        u32x4 t[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) t[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, mine, min(s0 + s, p.kparts - 1) * part_stride, 16);      // sc1: past the L1
        if constexpr (KW == 1) {
#pragma unroll
          for (int s = 0; s < 8; s += 2)
            if (s0 + s < p.kparts) {
              const f32x4 lo = __builtin_bit_cast(f32x4, t[s]);
              const f32x4 hi = s0 + s + 1 < p.kparts ? __builtin_bit_cast(f32x4, t[s + 1]) : f32x4{0.f, 0.f, 0.f, 0.f};
              a += lo + hi;
            }
        } else {
#pragma unroll
          for (int s = 0; s < 8; ++s)
            if (s0 + s < p.kparts) a += __builtin_bit_cast(f32x4, t[s]);
        }
      }
    } else {
      a = *reinterpret_cast<const f32x4*>(&outi[((uct * MT + umt) * 64 + lane) * 4]);
    }
    const int col = utile * 16 + (lane & 15);
    const bool col_ok = col < p.N;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = a[r];
      if (WT == WFMT_FP8) x *= e_s[i];      // power-of-two column scale: exact
      if (ln) {
        const float2 st = *reinterpret_cast<const float2*>(&stat[(umt * 16 + g4 + r) * 2]);
        x = st.y * (x - st.x * e_u[i]);
      }
      if (p.bias) x += e_bias[i];
      if (p.act == 1) x = gelu_new_pl(x);
      if (p.res) x += e_res[i][r];
      v[r] = x;
      const int row = umt * 16 + g4 + r;
      if (col_ok && row < p.rows) p.y[(size_t)row * p.ldy + col] = x;
    }
    if (p.stats_out) {      // (mean, M2) of every row over this tile's 16 columns, for the folded LayerNorm of the consumer
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mean = row16_sum(v[r]) * 0.0625f;
        const float dlt = v[r] - mean;
        const float q = row16_sum(dlt * dlt);
        if ((lane & 15) == 0) *reinterpret_cast<float2*>(&p.stats_out[((size_t)utile * R + umt * 16 + g4 + r) * 2]) = make_float2(mean, q);
      }
    }
  }
#ifdef PL_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PLS(10); PLR(11);
#endif
}

// ---- host side ----
static std::atomic<int> g_pl_min_rows{17};
void set_decode_plane_rows(int min_rows) { g_pl_min_rows.store(min_rows == 0 ? 17 : min_rows); }
int get_decode_plane_rows() { return g_pl_min_rows.load(); }

static int g_pl_ct_override = 0;      // tools/gemv_pl_probe.hip: force the column tiles per workgroup (0 = the plan below)
void gemv_pl_set_ct_override(int ct) { g_pl_ct_override = ct; }
static int pl_env_ct() {
  static const int v = [] { const char* e = getenv("IDXTTS_PL_CT"); return e ? atoi(e) : 0; }();
  return g_pl_ct_override ? g_pl_ct_override : v;
}

// Column tiles per workgroup (CT; K slices per workgroup = 8 / CT; K parts across workgroups follow).  Measured per shape and row
// count with tools/gemv_pl_probe.hip (profiles/README.md "Round 4"): CT = 4 (four K parts at K = 1280) unless the launch would not
// fit the chip in one round at that LDS footprint (-> 8: a third of the LDS); K = 5120 stays at 4 (the merge of 32 parts costs more
// than a second round).  CT = 4 and CT = 8 add the K groups in the same order (see the merge), so the choice never shows in a result.
void gemv_pl_plan(int N, int K, int rows, int* ct_out, int* kparts_out) {
  const int MT = cdiv(rows, 16), KC = cdiv(K, 32), ntiles = cdiv(N, 16);
  auto kparts = [&](int ct) { return cdiv(KC, (8 / ct) * PL_CPS); };
  auto fits = [&](int ct) { return pl_lds_bytes(MT, ct, true) <= 159 * 1024 && kparts(ct) <= 32; };
  int ct = 4;
  if (KC >= 128) ct = 4;
  else if (fits(4)) {
    const int per_cu = std::max(1, (160 * 1024) / pl_lds_bytes(MT, 4, kparts(4) == 1));
    if (cdiv(ntiles, 4) * kparts(4) > 256 * per_cu) ct = 8;
  }
  const int forced = pl_env_ct();
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) ct = forced;
  while (ct < 8 && !fits(ct)) ct *= 2;
  while (ct > 1 && kparts(ct) > 32) ct /= 2;
  *ct_out = ct;
  *kparts_out = kparts(ct);
}

size_t gemv_pl_slab_floats(int N, int K, int rows) {
  int ct, kp;
  gemv_pl_plan(N, K, rows, &ct, &kp);
  return kp > 1 ? (size_t)kp * cdiv(N, 16) * cdiv(rows, 16) * 256 : 0;
}

template <int MT, int CT, int WT>
static int pl_launch(const GemvPLP& p, dim3 grid, hipStream_t stream) {
  const int lds = pl_lds_bytes(MT, CT, 8 / CT > 1 || p.kparts == 1);
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_pl_kernel<MT, CT, WT>), hipFuncAttributeMaxDynamicSharedMemorySize, pl_lds_bytes(MT, CT, true)));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemv_pl_kernel<MT, CT, WT>), grid, dim3(512), lds, stream, p);
  return 0;
}

template <int MT, int WT>
static int pl_launch_ct(int ct, const GemvPLP& p, dim3 grid, hipStream_t stream) {
  switch (ct) {
    case 1: if constexpr (pl_lds_bytes(MT, 1, true) <= 159 * 1024) return pl_launch<MT, 1, WT>(p, grid, stream); break;
    case 2: if constexpr (pl_lds_bytes(MT, 2, true) <= 159 * 1024) return pl_launch<MT, 2, WT>(p, grid, stream); break;
    case 4: if constexpr (pl_lds_bytes(MT, 4, true) <= 159 * 1024) return pl_launch<MT, 4, WT>(p, grid, stream); break;
    case 8: return pl_launch<MT, 8, WT>(p, grid, stream);
  }
  IDX_FAIL("gemv_pl: no kernel for this geometry");
}

int gemv_pl_forward(const Gemv32Weights& w, const GemvPLArgs& a, hipStream_t stream) {
  IDX_CHECK(w.wp && a.x && a.y, "null pointer");
  IDX_CHECK(a.rows > 0 && a.rows <= 64, "1..64 rows");
  IDX_CHECK(w.fmt == WFMT_BF16 || (w.fmt == WFMT_FP8 && w.wscale), "weight format (bf16 / fp8 streams)");
  IDX_CHECK((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.ldx & 3) == 0 && (w.K & 3) == 0 && a.ldx >= w.K && (reinterpret_cast<uintptr_t>(w.wp) & 15) == 0,
            "alignment: x rows in 16-byte units");
  if (a.colsum) IDX_CHECK(a.stats_in && a.stats_tiles > 0 && a.stats_tiles * 16 == w.K, "folded LayerNorm needs the producer's row statistics per 16 columns of K");
  if (a.stats_out) IDX_CHECK(w.N % 16 == 0, "row statistics need N % 16 == 0");
  IDX_CHECK(a.ldy >= w.N, "ldy");
  GemvPLP p;
  p.x = a.x; p.ldx = a.ldx; p.wp = w.wp; p.wscale = w.wscale; p.bias = a.bias; p.colsum = a.colsum; p.ln_eps = a.ln_eps;
  p.stats_in = a.stats_in; p.stats_tiles = a.stats_tiles; p.res = a.res; p.y = a.y; p.ldy = a.ldy; p.stats_out = a.stats_out;
  p.slab = a.slab; p.cnt = a.counters; p.rows = a.rows; p.N = w.N; p.K = w.K; p.KC = cdiv(w.K, 32); p.ntiles = cdiv(w.N, 16); p.act = a.act;
  int ct = 4;
  gemv_pl_plan(w.N, w.K, a.rows, &ct, &p.kparts);
  const int MT = cdiv(a.rows, 16);
  if (p.kparts > 1) IDX_CHECK(a.slab && a.counters, "this shape splits K across workgroups: slab and counters needed");
  const size_t slab_bytes = (size_t)p.kparts * p.ntiles * MT * 1024;
  IDX_CHECK(slab_bytes < ((size_t)1 << 31), "slab size");
#ifdef PL_STAMPS
  p.stamps = g_pl_stamps;
#endif
  p.slab_bytes = p.kparts > 1 ? (int)slab_bytes : 0;
  if (p.kparts <= 1) p.slab = nullptr;
  dim3 grid(cdiv(p.ntiles, ct), p.kparts);
  const double flops = 2.0 * a.rows * (double)w.N * w.K;
  const double bytes = (double)wfmt_bytes(w.fmt) * w.N * w.K + 4.0 * a.rows * (double)w.K + 4.0 * a.rows * w.N * (a.res ? 2.0 : 1.0);
  static const int cat = prof_register("gemv_pl_kernel");
  ProfScope prof(cat, stream, flops, bytes);
  int rc;
#define PL_MT(MTV) (w.fmt == WFMT_FP8 ? pl_launch_ct<MTV, WFMT_FP8>(ct, p, grid, stream) : pl_launch_ct<MTV, WFMT_BF16>(ct, p, grid, stream))
  if (MT == 1) rc = PL_MT(1); else if (MT == 2) rc = PL_MT(2); else if (MT == 3) rc = PL_MT(3); else rc = PL_MT(4);
#undef PL_MT
  if (rc) return rc;
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
