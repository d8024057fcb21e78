// Token-major GEMM on the bf16 matrix core with fp32-class accuracy ("split-bf16", 3 products).
//
// Same contract as gemm.hip (Y = epi(X W^T + b), conv taps, paired gates, row masks) for the compute-bound
// shapes of the hot path: the s2mel DiT / WaveNet linears and convolutions (diffusion_transformer.py:213-252,
// gpt_fast/model.py:270-319, wavenet.py:149-161) and the GPT latent pass (model_v2.py:673-723).
//
// Why: gfx950 has no TF32/xf32; its f32-input MFMA runs at the VALU rate (157 TFLOP/s) while
// v_mfma_f32_32x32x16_bf16 runs 16x faster.  Every fp32 operand is written as hi + lo with hi = bf16(x),
// lo = bf16(x - hi) (16 significant bits), and x*w ~= hi*hi' + hi*lo' + lo*hi' is accumulated in the fp32
// accumulator: three bf16 MFMAs per product = 5.3x the f32 MFMA rate at a relative product error of ~2^-16
// (vs 2^-24 for fp32), far inside the path's stated tolerance (mel L1 <= 1e-3) -- measured in the parity tests.
// The KV-cached greedy decode stays on exact-fp32 kernels (token indices must be bit-exact).
//
// Structure: 128x128 tile / 256 threads (2x2 waves, 2x2 tiles of 32x32 per wave), K stepped 32 at a time.
// Weights are split and laid out at context creation exactly as the LDS image ([hl][128 rows][32 k + 8 pad] bf16,
// 20 KiB per (n-block, k-step)): their tile load is a linear 16-B/lane copy.  Activations are fp32 in HBM; the tile
// loader converts them to hi/lo on the way into LDS (v_cvt_pk_bf16_f32).  80-byte LDS rows make every
// ds_read_b128 fragment read conflict-free (rows r and r+4 of an unpadded 64-byte row would share banks).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "gemm_common.h"
#include "prof.h"

namespace idxtts {

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BROW = 40;                       // bf16 elements per LDS row (32 + 8 pad) = 80 bytes
constexpr int TILE_HALF = 128 * BROW;          // elements of one [128][40] image (hi or lo)
constexpr int WTILE_BYTES = 2 * TILE_HALF * 2; // 20480 bytes per packed (n-block, k-step) weight tile

static inline uint16_t f2bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// the pack holds two images: the padded LDS-image tiles of the register-staged kernels below, then the
// [hl][K/16][Npad][16] planes of the LDS-DMA kernel (gemm_bf16x3_v2.hip)
size_t linear_planes_bytes(int N, int K);
void pack_linear_planes(void* dst, const float* w, int N, int K);
bool gemm_bf16x3_uses_v2(const LinearWeights& w, const GemmArgs& a);
int gemm_bf16x3_v2_forward(GemmKP p, const void* wplanes, const LinearWeights& w, const GemmArgs& a, hipStream_t stream, double flops,
                           double bytes);
bool gemm_bf16x3_uses_v2(const LinearWeights& w, const GemmArgs& a) {
  // (M of a few hundred rows upwards: the 128 x 128 geometry of the LDS-DMA kernel takes what the 256 x 256 tiles would under-fill)
  return w.wp16 && a.M >= 256 && w.N >= 96 && w.K % 16 == 0 && (a.taps <= 1 || (w.K / a.taps) % 16 == 0);
}
static size_t tiles_bytes(int N, int K) { return (size_t)cdiv(N, 128) * cdiv(K, 32) * WTILE_BYTES; }
size_t linear_bf16x3_packed_bytes(int N, int K) { return tiles_bytes(N, K) + linear_planes_bytes(N, K); }

// w: [N][K] fp32 -> [N/128][K/32][hl][128][40] bf16
void pack_linear_bf16x3(void* dst, const float* w, int N, int K) {
  uint16_t* o = static_cast<uint16_t*>(dst);
  const int NB = cdiv(N, 128), KS = cdiv(K, 32);
  std::memset(o, 0, tiles_bytes(N, K));
  pack_linear_planes(static_cast<char*>(dst) + tiles_bytes(N, K), w, N, K);
  for (int nb = 0; nb < NB; ++nb)
    for (int ks = 0; ks < KS; ++ks) {
      uint16_t* tile = o + ((size_t)nb * KS + ks) * (2 * TILE_HALF);
      for (int r = 0; r < 128; ++r) {
        const int n = nb * 128 + r;
        if (n >= N) break;
        for (int kk = 0; kk < 32; ++kk) {
          const int k = ks * 32 + kk;
          if (k >= K) break;
          const float x = w[(size_t)n * K + k];
          const uint16_t hi = f2bf(x);
          tile[r * BROW + kk] = hi;
          tile[TILE_HALF + r * BROW + kk] = f2bf(x - bf2f(hi));
        }
      }
    }
}

__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const GemmKP p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 sm16[];
  __bf16* As = sm16;                        // [2 bufs][hl][128][40]
  __bf16* Bs = sm16 + 2 * 2 * TILE_HALF;    // [2 bufs][hl][128][40]

  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  int bn, bm;
  if (p.n_fast) { const int bml = q / p.nblocks; bn = q - bml * p.nblocks; bm = bml * 8 + xcd; }
  else { bn = q / p.mt8; bm = (q - bn * p.mt8) * 8 + xcd; }
  if (bm >= p.mtiles) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int ksteps = (p.K + 31) >> 5;

  f32x4 xr[4];
  f32x4 wr[5];
  int seq_base[4], seq_t[4], seq_n[4];
  if (p.taps > 1) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int m = bm * 128 + ((tid + 256 * l) >> 3);
      const int sb = m / p.seq_len;
      seq_base[l] = sb * p.seq_len;
      seq_t[l] = m - sb * p.seq_len;
      seq_n[l] = (p.row_len && m < p.M) ? min(p.row_len[sb], p.seq_len) : p.seq_len;
    }
  }
  const char* wtile0 = reinterpret_cast<const char*>(p.wp) + (size_t)bn * ksteps * WTILE_BYTES;
  auto load_tiles = [&](int kstep) {
    int tap = 0, kk0 = kstep * 32;
    if (p.taps > 1) { tap = kk0 / p.kc; kk0 -= tap * p.kc; }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 3, q8 = idx & 7;
      const int m = bm * 128 + row, k = kstep * 32 + q8 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p.taps > 1) {
        int t = seq_t[l] + tap * p.dil - p.pad_left;
        if (p.pad_mode == 1) { t = t < 0 ? -t : t; t = t >= seq_n[l] ? 2 * (seq_n[l] - 1) - t : t; }
        if (m < p.M && k < p.K && t >= 0 && t < seq_n[l])
          v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(seq_base[l] + t) * p.ldx + kk0 + q8 * 4);
      } else if (m < p.M && k < p.K) {
        v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + k);
      }
      xr[l] = v;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wtile0 + (size_t)kstep * WTILE_BYTES);
#pragma unroll
    for (int l = 0; l < 5; ++l) wr[l] = wsrc[tid + 256 * l];       // 1280 x 16 B = 20 KiB, linear
  };
  auto store_tiles = [&](int buf) {
    __bf16* a_hi = As + buf * 2 * TILE_HALF;
    __bf16* a_lo = a_hi + TILE_HALF;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 3, q8 = idx & 7;
      const bf16x4 hi = __builtin_convertvector(xr[l], bf16x4);
      const f32x4 back = __builtin_convertvector(hi, f32x4);
      const bf16x4 lo = __builtin_convertvector(xr[l] - back, bf16x4);
      *reinterpret_cast<bf16x4*>(a_hi + row * BROW + q8 * 4) = hi;
      *reinterpret_cast<bf16x4*>(a_lo + row * BROW + q8 * 4) = lo;
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(Bs + buf * 2 * TILE_HALF);
#pragma unroll
    for (int l = 0; l < 5; ++l) wdst[tid + 256 * l] = wr[l];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const bool has_next = ks + 1 < ksteps;
    if (has_next) load_tiles(ks + 1);
    const __bf16* a_hi = As + (ks & 1) * 2 * TILE_HALF;
    const __bf16* b_hi = Bs + (ks & 1) * 2 * TILE_HALF;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ar = (wm * 64 + t * 32 + j) * BROW + s * 16 + h * 8;
        const int br = (wn * 64 + t * 32 + j) * BROW + s * 16 + h * 8;
        ah[t] = *reinterpret_cast<const bf16x8*>(a_hi + ar);
        al[t] = *reinterpret_cast<const bf16x8*>(a_hi + TILE_HALF + ar);
        bh[t] = *reinterpret_cast<const bf16x8*>(b_hi + br);
        bl[t] = *reinterpret_cast<const bf16x8*>(b_hi + TILE_HALF + br);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    if (has_next) store_tiles((ks + 1) & 1);
    __syncthreads();
  }
  gemm_epilogue(p, acc, bm, bn, wm, wn, h, j);
}

// ---------------------------------------------------------------------------------------------------------
// Large-M variant: 256 x (64*TN) workgroup tile, 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x (32*TN).
// The 128x128 kernel above needs 48 B/clk/CU of operand traffic at the bf16 MFMA rate (measured: a CU sustains ~13,
// L2 hit rate 66 %); doubling the rows a weight tile is applied to halves the weight traffic per flop and doubles the
// MFMA work per barrier (one workgroup per CU, two waves per SIMD).
template <int TN>
__global__ __launch_bounds__(512) void gemm_bf16x3_big_kernel(const GemmKP p) {
  constexpr int BN = 64 * TN;                 // columns per workgroup
  constexpr int NB128 = BN / 128 > 0 ? BN / 128 : 1;   // packed 128-column weight tiles per k-step
  constexpr int A_HALF = 256 * BROW;          // elements of one [256][40] activation image
  constexpr int B_HALF = TILE_HALF;           // [128][40] per packed weight tile (hi or lo)
  constexpr int A_STAGE = 2 * A_HALF, B_STAGE = NB128 * 2 * B_HALF;
  constexpr int NWL = (NB128 * 1280 + 511) / 512;      // 16-byte weight loads per thread per k-step
  static_assert(BN == 128 || BN == 256, "tile");
  extern __shared__ __attribute__((aligned(16))) __bf16 sm16[];
  __bf16* As = sm16;                          // [2][hl][256][40]
  __bf16* Bs = sm16 + 2 * A_STAGE;            // [2][NB128][hl][128][40]

  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  int bn, bm;
  if (p.n_fast) { const int bml = q / p.nblocks; bn = q - bml * p.nblocks; bm = bml * 8 + xcd; }
  else { bn = q / p.mt8; bm = (q - bn * p.mt8) * 8 + xcd; }
  if (bm >= p.mtiles) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int ksteps = (p.K + 31) >> 5;

  f32x4 xr[4];
  f32x4 wr[NWL];
  int seq_base[4], seq_t[4], seq_n[4];
  if (p.taps > 1) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int m = bm * 256 + ((tid + 512 * l) >> 3);
      const int sb = m / p.seq_len;
      seq_base[l] = sb * p.seq_len;
      seq_t[l] = m - sb * p.seq_len;
      seq_n[l] = (p.row_len && m < p.M) ? min(p.row_len[sb], p.seq_len) : p.seq_len;
    }
  }
  const int n128 = cdiv_dev(p.N, 128);
  auto load_tiles = [&](int kstep) {
    int tap = 0, kk0 = kstep * 32;
    if (p.taps > 1) { tap = kk0 / p.kc; kk0 -= tap * p.kc; }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 512 * l;
      const int row = idx >> 3, q8 = idx & 7;
      const int m = bm * 256 + row, k = kstep * 32 + q8 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p.taps > 1) {
        int t = seq_t[l] + tap * p.dil - p.pad_left;
        if (p.pad_mode == 1) { t = t < 0 ? -t : t; t = t >= seq_n[l] ? 2 * (seq_n[l] - 1) - t : t; }
        if (m < p.M && k < p.K && t >= 0 && t < seq_n[l])
          v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(seq_base[l] + t) * p.ldx + kk0 + q8 * 4);
      } else if (m < p.M && k < p.K) {
        v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + k);
      }
      xr[l] = v;
    }
#pragma unroll
    for (int l = 0; l < NWL; ++l) {
      const int idx = tid + 512 * l;                 // 16-byte unit inside this k-step's weight tiles
      const int sub = idx / 1280, off = idx - sub * 1280;
      const int nb = bn * NB128 + sub;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NB128 * 1280 && nb < n128)
        v = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.wp) + ((size_t)nb * ksteps + kstep) * WTILE_BYTES)[off];
      wr[l] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    __bf16* a_hi = As + buf * A_STAGE;
    __bf16* a_lo = a_hi + A_HALF;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 512 * l;
      const int row = idx >> 3, q8 = idx & 7;
      const bf16x4 hi = __builtin_convertvector(xr[l], bf16x4);
      const f32x4 back = __builtin_convertvector(hi, f32x4);
      const bf16x4 lo = __builtin_convertvector(xr[l] - back, bf16x4);
      *reinterpret_cast<bf16x4*>(a_hi + row * BROW + q8 * 4) = hi;
      *reinterpret_cast<bf16x4*>(a_lo + row * BROW + q8 * 4) = lo;
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(Bs + buf * B_STAGE);
#pragma unroll
    for (int l = 0; l < NWL; ++l) {
      const int idx = tid + 512 * l;
      if (idx < NB128 * 1280) wdst[idx] = wr[l];
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const bool has_next = ks + 1 < ksteps;
    if (has_next) load_tiles(ks + 1);
    const __bf16* a_hi = As + (ks & 1) * A_STAGE;
    const __bf16* b_st = Bs + (ks & 1) * B_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ar = (wm * 64 + t * 32 + j) * BROW + s * 16 + h * 8;
        ah[t] = *reinterpret_cast<const bf16x8*>(a_hi + ar);
        al[t] = *reinterpret_cast<const bf16x8*>(a_hi + A_HALF + ar);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int col = wn * 32 * TN + t * 32 + j;                 // column inside the workgroup tile
        const __bf16* bt = b_st + (col >> 7) * 2 * B_HALF + (col & 127) * BROW + s * 16 + h * 8;
        bh[t] = *reinterpret_cast<const bf16x8*>(bt);
        bl[t] = *reinterpret_cast<const bf16x8*>(bt + B_HALF);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    if (has_next) store_tiles((ks + 1) & 1);
    __syncthreads();
  }
  gemm_epilogue_t<2, TN>(p, acc, bm * 256 + wm * 64, bn * BN + wn * 32 * TN, h, j);
}

template <int TN>
static int launch_big(GemmKP p, const LinearWeights& w, const GemmArgs& a, hipStream_t stream, double flops, double bytes) {
  constexpr int BN = 64 * TN, NB128 = BN / 128;
  p.mtiles = cdiv(a.M, 256);
  p.mt8 = cdiv(p.mtiles, 8);
  p.nblocks = cdiv(w.N, BN);
  const int64_t grid = (int64_t)8 * p.nblocks * p.mt8;
  IDX_CHECK(grid < (1ll << 31), "grid size");
  constexpr size_t lds = (size_t)(2 * 2 * 256 * BROW + 2 * NB128 * 2 * TILE_HALF) * sizeof(__bf16);
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_big_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  static const int cat = prof_register(TN == 2 ? "gemm_bf16x3_big_kernel<2>" : "gemm_bf16x3_big_kernel<4>");
  ProfScope prof(cat, stream, flops, bytes);
  hipLaunchKernelGGL(gemm_bf16x3_big_kernel<TN>, dim3((unsigned)grid), dim3(512), lds, stream, p);
  IDX_LAUNCH_CHECK();
  return 0;
}

int gemm_bf16x3_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream) {
  IDX_CHECK(w.wp16 && (a.x || a.x_planes) && (a.y || a.y_planes), "null pointer (split-bf16 weights not packed?)");
  if (a.M == 0) return 0;
  IDX_CHECK(a.M > 0 && w.N > 0 && w.K > 0, "bad shape");
  IDX_CHECK((w.K & 3) == 0 && (a.ldx & 3) == 0, "K and ldx must be multiples of 4");
  IDX_CHECK((reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "x must be 16-byte aligned");
  if (a.x_planes || a.y_planes || !a.x || !a.y || a.rope) IDX_CHECK(gemm_bf16x3_uses_v2(w, a), "operand planes are only understood by the LDS-DMA kernel (shape not eligible)");
  if (a.act == ACT_SWIGLU || a.act == ACT_GATE) IDX_CHECK((w.N & 63) == 0, "paired activations need N % 64 == 0");
  if (a.taps > 1) {
    IDX_CHECK(a.seq_len > 0 && a.M % a.seq_len == 0 && w.K % a.taps == 0 && ((w.K / a.taps) & 31) == 0, "conv mode shape");
    if (a.pad_mode == 1) IDX_CHECK(a.seq_len > (a.taps - 1) * a.dil, "reflect pad needs seq_len > halo");
  }
  if (a.row_len) IDX_CHECK(a.seq_len > 0, "row_len needs seq_len");
  GemmKP p;
  p.x = a.x; p.wp = reinterpret_cast<const float*>(w.wp16); p.bias = w.bias; p.res = a.res; p.y = a.y; p.y_hi = p.y_lo = nullptr; p.rope = nullptr; p.rope_T = 1; p.rope_cols = 0;
  p.M = a.M; p.N = w.N; p.K = w.K; p.ldx = a.ldx; p.ldy = a.ldy; p.ldr = a.ldr;
  p.kc16 = cdiv(w.K, 16);
  p.mtiles = cdiv(a.M, 128);
  p.mt8 = cdiv(p.mtiles, 8);
  p.act = a.act; p.out_scale = a.out_scale;
  p.taps = a.taps; p.kc = w.K / std::max(1, a.taps); p.seq_len = a.seq_len > 0 ? a.seq_len : 1; p.dil = a.dil; p.pad_left = a.pad_left;
  p.pad_mode = a.pad_mode; p.row_len = a.row_len;
  p.ksplit = 1; p.ksteps_per_split = 0;
  IDX_CHECK(a.ksplit <= 1, "split-K is a feature of the exact-fp32 kernel");
  const int nblocks = cdiv(w.N, 128);
  p.nblocks = nblocks;
  p.n_fast = ((double)w.N * w.K * 4.0 <= 8.0 * 1024 * 1024) && (a.M > w.N) ? 1 : 0;
  const int64_t grid = (int64_t)8 * nblocks * p.mt8;
  IDX_CHECK(grid < (1ll << 31), "grid size");
  const double flops = 2.0 * a.M * (double)w.N * w.K;
  const double bytes = 4.0 * ((double)a.M * w.K + (double)w.N * w.K + (double)a.M * w.N * (a.res ? 2.0 : 1.0));
  if (gemm_bf16x3_uses_v2(w, a))
    return gemm_bf16x3_v2_forward(p, static_cast<const char*>(w.wp16) + tiles_bytes(w.N, w.K), w, a, stream, flops, bytes);
  if (a.M >= 4096) {      // shapes the LDS-DMA kernel does not take (N < 192, K % 16 != 0): the register-staged 256-row tiles
    const bool paired = a.act == ACT_SWIGLU || a.act == ACT_GATE;
    if (w.N >= 256 && !paired) return launch_big<4>(p, w, a, stream, flops, bytes);
    return launch_big<2>(p, w, a, stream, flops, bytes);
  }
  constexpr size_t lds = (size_t)(2 * 2 * 2 * TILE_HALF) * sizeof(__bf16);
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  static const int cat = prof_register("gemm_bf16x3_kernel");
  ProfScope prof(cat, stream, flops, bytes);
  hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3((unsigned)grid), dim3(256), lds, stream, p);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
