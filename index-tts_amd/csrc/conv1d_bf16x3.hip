// Split-bf16 variant of the channels-first implicit-GEMM Conv1d (conv1d.hip) for the wide vocoder layers (M > 96 rows:
// conv_pre, the first four upsamplers and AMP-block stages of BigVGAN, bigvgan.py:362-379): same tiling, same epilogue,
// but the contraction runs on v_mfma_f32_32x32x16_bf16 with every fp32 operand written as hi + lo (bf16 each) and three
// MFMAs per product (hi*hi + hi*lo + lo*hi, fp32 accumulation; relative product error ~2^-16).  12 bf16 MFMAs (384
// cycles) per (16-channel chunk, tap) and 64 x 64 wave tile instead of 32 f32 MFMAs (2048 cycles).
//
// What changes against the fp32 kernel is the x tile: the bf16 MFMA wants the 8 input channels of a k-slot group packed
// in ONE lane, so the tile is stored time-major in LDS, [t][16 ci] bf16 (32-byte rows, hi and lo planes).  The loader
// assigns a lane 4 adjacent channels x strided time steps (coalesced 4-byte global loads along t), packs the 4 channels
// to 8 bytes and writes row t: the 64 lanes of a store cover 16 complete rows = 512 contiguous bytes (conflict-free).
// A tap is a shifted ROW, so every dilation / tap offset is a plain ds_read_b128 (16-byte slot swizzled by row bit 3:
// conflict-free).  Weights are permuted once at load from the fp32 pack into per-sub-tile [hl][g2][i32][8] bf16 images
// (same 2 KiB per sub-tile and tap).
#include <cstdlib>
#include <cstring>

#include "conv1d.h"
#include "conv_epilogue.h"
#include <string>

#include "prof.h"

namespace idxtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static inline uint16_t f2bf_c(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f_c(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// fp32 pack [.. sub-tile ..][g2][h2][i32][4] (ci = 8g + 4h + e)  ->  [.. sub-tile ..][hl][g2][i32][8] bf16 (k-slot 4h + e)
void pack_conv_bf16x3(void* dst, const float* packed_f32, size_t n_subtiles) {
  uint16_t* o = static_cast<uint16_t*>(dst);
  for (size_t s = 0; s < n_subtiles; ++s) {
    const float* src = packed_f32 + s * CONV_SUB;
    uint16_t* hi = o + s * 1024;      // 2 planes x 512 bf16
    uint16_t* lo = hi + 512;
    for (int g = 0; g < 2; ++g)
      for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 32; ++i)
          for (int e = 0; e < 4; ++e) {
            const float x = src[((g * 2 + h) * 32 + i) * 4 + e];
            const uint16_t xh = f2bf_c(x);
            const int d = (g * 32 + i) * 8 + 4 * h + e;
            hi[d] = xh;
            lo[d] = f2bf_c(x - bf2f_c(xh));
          }
  }
}

struct ConvKP16 {
  const float* x;
  const void* wp16;
  const float* bias;
  const float* res;
  float* y;
  int Cin, M, T;
  int K, dil, pad_left, pad_mode;
  int nchunk, mt32;
  int ups_log2;
  int xt;                 // x tile rows (BN + (K-1)*dil)
  int ntiles_row, ntiles, nt8;
  int mblocks, walk;      // walk 0: row blocks slowest (an XCD sweeps the time tiles of ONE row block: its weight slice stays in L2, x is re-read per row block)
                          //      1: row blocks fastest (the row blocks of one time tile run side by side on an XCD: x is read from HBM once, weights stream through L2)
  float scale;
  int accum;
  const int* lens; int len_mul_out;
  int w_bytes;            // size of the weight pack (buffer resource bound)
  int w_mt_stride;        // bytes between two 32-row sub-tiles of the pack: nchunk * K * 2048
};

constexpr int XROW_B = 32;            // bytes per LDS x row (16 ci bf16)

// K1: a 1-tap convolution (every step is a chunk's first and last tap).  Compiled apart because the order of a step's requests
// differs: with K > 1 the next chunk's x rows are requested BEHIND the step's weight tile and stored at least one step later, so the
// counted wait in front of their store (vmcnt = the weight requests issued since) never drains the weight tiles in flight, and the x
// rows get a whole step to arrive; with K = 1 they are stored in the step that requested them.
template <int TM, int TN, int WGM, int WGN, bool K1>
__device__ __forceinline__ void conv1d_bf16x3_body(const ConvKP16& p) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int NSUB = TM * WGM;
  constexpr int NXJ = (BN + CONV_MAX_HALO + 63) / 64;    // time steps per lane in the x loader
  constexpr int NW4 = NSUB * 128;                        // 16-byte units per weight tile (2 KiB per sub-tile)
  constexpr int NWL = (NW4 + 255) / 256;
  constexpr int XT_MAX = BN + CONV_MAX_HALO;
  constexpr int XPLANE = XT_MAX * XROW_B;                // bytes of one x plane
  static_assert(WGM * WGN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem16[];
  // bytes of a weight ring slot: what NWL DMA instructions of 256 lanes x 16 B write; the one-sub-tile configuration (2 KiB per step)
  // lets only waves 0 and 1 request, so that its 74 KiB of x planes + ring stay under half a CU's LDS (two workgroups per CU)
  constexpr int WSLOT = NSUB == 1 ? 2048 : NWL * 4096;
  // a tile is requested DIST steps ahead of its use.  (3 steps ahead for the narrow tiles was measured: the fourth ring slot took the
  // 64 x 256 configuration from three workgroups per CU to two: 188 -> 158 TF-eq.)
  constexpr int DIST = 2;
  constexpr int NRING = DIST + 1;
  char* Ws = smem16;                           // [NRING][WSLOT]: a step's [NSUB][hl][g2][i32][8] bf16 tile, copied linearly by LDS-DMA
  char* Xs = smem16 + NRING * WSLOT;           // [2][hl][XT_MAX][16 ci] bf16

  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  const int m_blk = p.walk ? q % p.mblocks : q / p.nt8;
  const int n_idx = (p.walk ? q / p.mblocks : q - m_blk * p.nt8) * 8 + xcd;
  if (n_idx >= p.ntiles) return;
  const int b = n_idx / p.ntiles_row;
  const int t0 = (n_idx - b * p.ntiles_row) * BN;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  const int T = p.T, XT = p.xt;
  const float* xrow_base = p.x + (size_t)b * p.Cin * T;

  // x loader role: channels 4cg..4cg+3 of the chunk, tile rows tq + 64*jj
  const int cg = tid & 3, tq = tid >> 2;
  float xr[NXJ][4];

  // Every global load of the loop is issued unconditionally, at a clamped address, and what must read as zero (padding in time,
  // channels past Cin, weight sub-tiles past M) is zeroed when the registers are STORED to LDS: with a branch around each load
  // ("load or zero") the compiler cannot count the loads in flight and waited vmcnt(0) in front of every ds_write -- which drained
  // the weight tile requested two steps ahead and the next chunk's x rows with it.
  auto x_time = [&](int jj, bool& tok) {
    const int c = tq + 64 * jj;
    int t = t0 - p.pad_left + c;
    if (p.pad_mode == PAD_REFLECT) {
      t = t < 0 ? -t : t;
      t = t >= T ? 2 * (T - 1) - t : t;
    }
    tok = c < XT && t >= 0 && t < T;
    return min(max(t, 0), T - 1);
  };
  auto load_x = [&](int chunk) {
#pragma unroll
    for (int jj = 0; jj < NXJ; ++jj) {
      bool tok;
      const int t = x_time(jj, tok);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ci = min(chunk * CONV_KC + cg * 4 + e, p.Cin - 1);
        xr[jj][e] = xrow_base[(size_t)ci * T + t];
      }
    }
  };
  auto store_x = [&](int buf, int chunk) {
    char* dst = Xs + buf * 2 * XPLANE;
#pragma unroll
    for (int jj = 0; jj < NXJ; ++jj) {
      const int c = tq + 64 * jj;
      bool tok;
      (void)x_time(jj, tok);
      {      // rows XT .. XT_MAX - 1 (c <= 64 NXJ - 1 = XT_MAX - 1) are never read: written unconditionally
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (tok && chunk * CONV_KC + cg * 4 + e < p.Cin) ? xr[jj][e] : 0.0f;
        bf16x4 hi, lo;
        split_bf16_x4(v, hi, lo);
        const int off = c * XROW_B + (((cg >> 1) ^ ((c >> 3) & 1)) << 4) + ((cg & 1) << 3);
        *reinterpret_cast<bf16x4*>(dst + off) = hi;
        *reinterpret_cast<bf16x4*>(dst + XPLANE + off) = lo;
      }
    }
  };
  // Weight tiles travel global -> LDS by DMA (buffer_load ... lds: no registers, no ds_write), two steps ahead of their use through a
  // ring of three slots.  A lane's source offset inside a step's tile never changes (sub-tile, 16-byte unit); the step selects the
  // tile through the scalar offset.  Sub-tiles past M read the last one again (the epilogue discards those rows).
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp16), 0, p.w_bytes, 0x00020000);
  int wvoff[NWL];
#pragma unroll
  for (int l = 0; l < NWL; ++l) {
    const int idx = min(tid + l * 256, NW4 - 1);
    wvoff[l] = min(m_blk * NSUB + (idx >> 7), p.mt32 - 1) * p.w_mt_stride + (idx & 127) * 16;
  }
  char* const lds_w = Ws + wave * 1024;
  auto request_w = [&](int slot, int chunk, int tap) {
    const int so = (chunk * p.K + tap) * 2048;
    if (NSUB == 1 && wave >= 2) return;      // (wave-uniform; waves 2 and 3 hold no weight request: their counted waits only see x loads)
#pragma unroll
    for (int l = 0; l < NWL; ++l)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(lds_w + slot * WSLOT + l * 4096), 16, wvoff[l], so, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  // Prefetch: a weight tile is requested DIST steps ahead of its use (a step is 12 MFMAs per wave, ~400 cycles: less than one L2 round
  // trip under load) into a ring of DIST + 1 slots.  The x rows of the next chunk are requested into registers at the top of the
  // chunk's first step, IN FRONT of that step's weight request -- the compiler's own wait in front of their store (last tap) then leaves
  // the youngest weight request in flight -- and stay in flight across DIST - 1 barriers: vmcnt completes in order, so a load can be
  // outstanding only while the tile a step waits for is older than it.  All waits on the weight ring are explicit and counted; the only
  // loads the compiler schedules waits for are the x registers.  (Requested at the END of the first step instead: 339 vs 345 TF-eq.)
#define CONV_WAITCNT(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | (((vm) >> 4) << 14) | 0x70)   /* vmcnt(vm) lgkmcnt(0) */
  constexpr int NXL = 4 * NXJ;                 // x loads of a chunk per lane
  constexpr int NWF = (DIST - 1) * NWL;        // weight requests younger than the tile a step waits for
  static_assert(NWF + NXL < 64, "vmcnt is 6 bits");
  const int total = p.nchunk * p.K;
  auto step_of = [&](int it, int& c, int& t) { c = it / p.K; t = it - c * p.K; };
  load_x(0);
#pragma unroll
  for (int d = 0; d < DIST; ++d) { int c1, t1; step_of(min(d, total - 1), c1, t1); request_w(d, c1, t1); }
  store_x(0, 0);
  CONV_WAITCNT(NWF);      // step 0's tile has landed; the x rows are stored
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int chunk = 0, tap = 0, rslot = 0;
  int fchunk = 0, ftap = 0;      // (chunk, tap) of step it + DIST - 1: advanced once more and requested at the top of step it
  for (int d = 1; d < DIST; ++d) { if (++ftap == p.K) { ftap = 0; ++fchunk; } }
  int x_age = DIST;              // steps since the x rows in flight were requested (>= DIST - 1: none may be outstanding)
  for (int it = 0; it < total; ++it) {
    int nchunk_i = chunk, ntap = tap + 1;
    if (ntap == p.K) { ntap = 0; nchunk_i = chunk + 1; }
    const bool has_next = it + 1 < total;
    const bool x_ahead = !K1 ? (tap == 0 && chunk + 1 < p.nchunk) : has_next;
    if (x_ahead) { load_x(chunk + 1); x_age = 0; }
    {      // step it + DIST's tile into the slot step it - 1 read (past the end: the last tile again, nobody reads it)
      if (++ftap == p.K) { ftap = 0; ++fchunk; }
      const bool past = it + DIST >= total;
      const int wslot = rslot == 0 ? NRING - 1 : rslot - 1;
      request_w(wslot, past ? p.nchunk - 1 : fchunk, past ? p.K - 1 : ftap);
    }
    {
      const char* xb = Xs + (chunk & 1) * 2 * XPLANE;
      const char* wb = Ws + rslot * WSLOT;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const char* wsub = wb + (wm * TM + mt) * 2048 + (h * 32 + j) * 16;
        ah[mt] = *reinterpret_cast<const bf16x8*>(wsub);
        al[mt] = *reinterpret_cast<const bf16x8*>(wsub + 1024);
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int row = (wn * TN + nt) * 32 + j + tap * p.dil;
        const int off = row * XROW_B + ((h ^ ((row >> 3) & 1)) << 4);
        bh[nt] = *reinterpret_cast<const bf16x8*>(xb + off);
        bl[nt] = *reinterpret_cast<const bf16x8*>(xb + XPLANE + off);
      }
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    if (has_next && ntap == 0) { store_x(nchunk_i & 1, nchunk_i); x_age = DIST; }
    // step it + 1's tile (requested at the top of step it + 1 - DIST) has landed when nothing older than the DIST - 1 younger weight
    // requests -- and than x rows requested less than DIST - 1 steps ago, which are younger than that tile too -- is in flight
    if (x_age < DIST - 1) CONV_WAITCNT(NWF + NXL); else CONV_WAITCNT(NWF);
    ++x_age;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    chunk = nchunk_i;
    tap = ntap;
    rslot = rslot == NRING - 1 ? 0 : rslot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the requests past the end are still writing LDS
#undef CONV_WAITCNT

  // ---- epilogue: bias, residual, scale, (accumulate), store: operands requested in batches ahead of the stores (conv_epilogue.h) ----
  conv_epilogue<TM, TN>(p, acc, m_blk * BM + wm * TM * 32, t0 + wn * TN * 32, b, T, h, j);
}

// (the body lives in a __device__ function: the host pass cannot instantiate a __global__ template whose body names device builtins)
template <int TM, int TN, int WGM, int WGN, bool K1>
__global__ __launch_bounds__(256, (TM == 3 ? 2 : 3)) void conv1d_bf16x3_kernel(const ConvKP16 p) {
  conv1d_bf16x3_body<TM, TN, WGM, WGN, K1>(p);
}

template <int TM, int TN, int WGM, int WGN, bool K1>
static int launch_conv16(const ConvWeights& w, const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, NSUB = TM * WGM;
  constexpr int XT_MAX = BN + CONV_MAX_HALO;
  IDX_CHECK(w.wp16 && a.x && a.y, "null pointer");
  ConvKP16 p;
  p.x = a.x; p.wp16 = w.wp16; p.bias = w.bias; p.res = a.res; p.y = a.y;
  p.Cin = w.Cin; p.M = w.M; p.T = a.T;
  p.K = w.K; p.dil = a.dil; p.pad_left = a.pad_left; p.pad_mode = a.pad_mode;
  p.nchunk = w.nchunk; p.mt32 = cdiv(w.M, CONV_MT);
  int ul = 0;
  while ((1 << ul) < w.ups) ++ul;
  IDX_CHECK((1 << ul) == w.ups, "transposed-conv stride must be a power of two");
  p.ups_log2 = ul;
  const int halo = (w.K - 1) * a.dil;
  IDX_CHECK(halo <= CONV_MAX_HALO, "(K-1)*dil exceeds CONV_MAX_HALO");
  p.xt = BN + halo;
  p.ntiles_row = cdiv(a.T, BN);
  p.ntiles = p.ntiles_row * a.B;
  p.nt8 = cdiv(p.ntiles, 8);
  p.scale = a.scale; p.accum = a.accum; p.lens = a.lens; p.len_mul_out = a.len_mul_out;
  const int mblocks = cdiv(w.M, BM);
  // Tile walk, per shape (FETCH_SIZE per launch under both walks: profiles/r04_conv_walk_pmc.txt): row blocks fastest -- x leaves HBM once,
  // the row blocks' weight slices stream through the XCD's L2 once per round of resident workgroups -- unless the layer's weights are so
  // large (> 20 MB: the first upsampler, the 768-channel k = 11 convolutions) that this stream outweighs the x re-reads it saves.
  // IDXTTS_CONV_WALK = 0 / 1 forces one walk for that comparison.
  static const int walk_env = [] { const char* e = getenv("IDXTTS_CONV_WALK"); return e ? atoi(e) : -1; }();
  const double weight_bytes = 4.0 * w.M * (double)w.Cin * w.K;
  p.mblocks = mblocks;
  p.walk = mblocks > 1 && (walk_env >= 0 ? walk_env == 1 : weight_bytes <= 20e6) ? 1 : 0;
  constexpr int NWL = (NSUB * 128 + 255) / 256;
  const size_t lds = (size_t)3 * (NSUB == 1 ? 2048 : NWL * 4096) + (size_t)2 * 2 * XT_MAX * XROW_B;
  const size_t w_bytes = (size_t)cdiv(w.M, CONV_MT) * w.nchunk * w.K * 2048;
  IDX_CHECK(w_bytes < (1ull << 31), "weight pack beyond the 2 GiB a buffer offset addresses");
  p.w_bytes = (int)w_bytes; p.w_mt_stride = w.nchunk * w.K * 2048;
  const int64_t grid = (int64_t)8 * mblocks * p.nt8;
  IDX_CHECK(grid > 0 && grid < (1ll << 31), "grid size");
  auto kern = conv1d_bf16x3_kernel<TM, TN, WGM, WGN, K1>;
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const double cout = (double)(w.M / w.ups), tout = (double)a.T * w.ups * a.B;
  const double taps = w.ups > 1 ? 2.0 : (double)w.K;
  const double flops = 2.0 * cout * w.Cin * taps * tout;
  const double bytes = 4.0 * ((double)a.B * w.Cin * a.T + cout * tout * (1.0 + (a.res ? 1.0 : 0.0) + (a.accum ? 1.0 : 0.0)) +
                              cout * w.Cin * (w.ups > 1 ? 2.0 * w.ups : (double)w.K));
  static const int cat = prof_register(("conv1d_bf16x3_kernel<" + std::to_string(TM) + ", " + std::to_string(TN) + ", " + std::to_string(WGM) + ", " + std::to_string(WGN) + (K1 ? ", true>" : ", false>")).c_str());
  ProfScope prof(cat, stream, flops, bytes);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
  IDX_LAUNCH_CHECK();
  return 0;
}

// same tile configurations as conv1d_forward
int conv1d_bf16x3_forward(const ConvWeights& w, const ConvArgs& a, hipStream_t stream) {
  if (w.K == 1) {
    if (w.M > 96) return launch_conv16<2, 2, 2, 2, true>(w, a, stream);
    if (w.M > 64) return launch_conv16<3, 2, 1, 4, true>(w, a, stream);
    if (w.M > 32) return launch_conv16<2, 2, 1, 4, true>(w, a, stream);
    return launch_conv16<1, 4, 1, 4, true>(w, a, stream);
  }
  if (w.M > 96) return launch_conv16<2, 2, 2, 2, false>(w, a, stream);   // 128 x 128
  if (w.M > 64) return launch_conv16<3, 2, 1, 4, false>(w, a, stream);   //  96 x 256
  if (w.M > 32) return launch_conv16<2, 2, 1, 4, false>(w, a, stream);   //  64 x 256
  return launch_conv16<1, 4, 1, 4, false>(w, a, stream);                 //  32 x 512
}

}  // namespace idxtts
