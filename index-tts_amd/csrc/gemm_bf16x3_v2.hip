// Split-bf16 token-major GEMM, LDS-DMA pipeline (256 x 256 tile): same contract and arithmetic as gemm_bf16x3.hip
// (Y = epi(X W^T + b), x*w ~= hi*hi' + hi*lo' + lo*hi' on v_mfma_f32_32x32x16_bf16, fp32 accumulation), different
// data movement.
//
// Why (profiles/r01 PMC, gemm_bf16x3_big_kernel<4> at M=50208 N=1024 K=2560): MFMA busy 30 %, waves parked 39 % of their
// lifetime on the s_waitcnt in front of the fp32 -> hi/lo conversion of the NEXT activation tile -- the register-staged
// loop can only prefetch one 72 KB K-step per CU and issues it as one burst (tools/l2_probe.hip: a CU moves 34 B/clk
// from L2 with 64 KB in flight, 18 B/clk with 32 KB, ~11 B/clk from HBM), and two register sets do not fit beside the
// 128 accumulator registers.  Here nothing on the operand path touches a VGPR or the VALU:
//   * both operands live in HBM already split, as planes [hl][K/16][rows][16] bf16 (a 16-k chunk of all rows is one
//     contiguous run): the weights are laid out like that at load, the activations by one streaming pre-pass
//     (split_planes_kernel; 8 B/element moved once per GEMM instead of a conversion in every column-block);
//   * tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (1 KiB per wave-instruction) into a 4-deep ring of 16-k stages
//     (32 KiB each); three stages are always in flight, retired with counted s_waitcnt vmcnt(8/4/0) and ONE raw
//     s_barrier per stage (RAW: wait, then barrier, then read; WAR: the stage overwritten after barrier i was last read
//     before it);
//   * LDS rows are 32 B (2 x 16-B units); unit' = unit ^ (row >> 3 & 1), applied on the DMA's SOURCE address and on the
//     ds_read_b128 address, makes every fragment read conflict-free without padding (the DMA destination is lane-linear).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "gemm_common.h"
#include "prof.h"

namespace idxtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static inline uint16_t f2bf_v2(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f_v2(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// ---- weight planes: [hl][K/16][Npad][16] bf16, Npad = N rounded up to 256 (zero rows), K rounded up to 16 ----
size_t linear_planes_bytes(int N, int K) { return (size_t)2 * cdiv(K, 16) * (cdiv(N, 256) * 256) * 16 * sizeof(uint16_t); }

void pack_linear_planes(void* dst, const float* w, int N, int K) {
  uint16_t* o = static_cast<uint16_t*>(dst);
  const int KC = cdiv(K, 16), NP = cdiv(N, 256) * 256;
  std::memset(o, 0, linear_planes_bytes(N, K));
  uint16_t* lo = o + (size_t)KC * NP * 16;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float x = w[(size_t)n * K + k];
      const uint16_t hi = f2bf_v2(x);
      const size_t idx = ((size_t)(k >> 4) * NP + n) * 16 + (k & 15);
      o[idx] = hi;
      lo[idx] = f2bf_v2(x - bf2f_v2(hi));
    }
}

// ---- activation pre-pass: x fp32 [M][ldx] -> planes [hl][K/16][M][16] bf16 ----
// Workgroup = 64 rows x 64 k.  Reads: 16 lanes per row (256 B = two full lines); writes: the four lanes of a 16-k chunk
// produce 32 B and the four rows of one wave-instruction are adjacent in the plane, so every store instruction fills
// four full 128-byte lines per plane.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ldx, int M, int K, __bf16* __restrict__ hi,
                                                           __bf16* __restrict__ lo) {
  const int k = blockIdx.y * 64 + (threadIdx.x & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = blockIdx.x * 64 + (threadIdx.x >> 4) + 16 * i;
    if (m >= M || k >= ((K + 15) & ~15)) continue;
    const f32x4 v = k < K ? *reinterpret_cast<const f32x4*>(x + (size_t)m * ldx + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x4 h, l;
    split_bf16_x4(v, h, l);
    const size_t o = ((size_t)(k >> 4) * M + m) * 16 + (k & 15);
    *reinterpret_cast<bf16x4*>(hi + o) = h;
    *reinterpret_cast<bf16x4*>(lo + o) = l;
  }
}

struct GemmV2P {
  GemmKP g;                      // shapes, epilogue operands, conv parameters (x unused)
  const __bf16* a_hi; const __bf16* a_lo;   // [K/16][a_rows][16]
  const __bf16* b_hi; const __bf16* b_lo;   // [K/16][npad][16]
  const __bf16* zeros;           // >= 32 bytes of zeros (rows outside M / outside the sequence)
  int a_rows, npad, nstages;
};

#define GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vm_lgkm0();
template <> __device__ __forceinline__ void wait_vm_lgkm0<0>() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm_lgkm0<4>() { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm_lgkm0<6>() { asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm_lgkm0<8>() { asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm_lgkm0<12>() { asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm_lgkm0<18>() { asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); }

// Workgroup = 4 x WNW waves, each a 64 x 128 output tile (2 x 4 MFMA tiles): 256 x (128 WNW) per workgroup.
//   WNW = 2: 512 threads, 32 KiB stages, 4-deep ring, one workgroup per CU;
//   WNW = 1: 256 threads, 24 KiB stages, 3-deep ring, TWO workgroups per CU -- the waves that share a SIMD then belong to
//            different workgroups with independent barriers, so one multiplies while the other waits / issues DMA.
template <int WNW, int NSTAGE>
__device__ __forceinline__ void gemm_v2_tile(const GemmV2P& q, const int bm, const int bn, const int tid) {
  constexpr int NWV = 4 * WNW;                 // waves
  constexpr int BN = 128 * WNW;
  constexpr int A_PLANE = 256 * 32, B_PLANE = BN * 32;
  constexpr int STAGE_BYTES = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int ARB = 8 / NWV, BRB = (BN / 32) / NWV;      // 32-row blocks of A / B a wave copies per plane and stage
  constexpr int PPW = 2 * (ARB + BRB);                     // DMA instructions per wave and stage
  const GemmKP& p = q.g;
  extern __shared__ __attribute__((aligned(1024))) char smv2[];

  const int wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave / WNW, wn = wave % WNW;

  // ---- DMA role of this lane: row (lane >> 1) of the wave's 32-row blocks, LDS unit lane & 1, source unit swizzled ----
  const int lrow = lane >> 1;
  const int dunit = (lane & 1) ^ ((lrow >> 3) & 1);        // block bases are multiples of 32: row bit 3 = lrow bit 3
  int a_m[ARB], seq_base[ARB], seq_t[ARB], seq_n[ARB];
#pragma unroll
  for (int r = 0; r < ARB; ++r) {
    a_m[r] = bm * 256 + 32 * (wave * ARB + r) + lrow;
    seq_base[r] = seq_t[r] = seq_n[r] = 0;
    if (p.taps > 1) {
      const int sb = a_m[r] / p.seq_len;
      seq_base[r] = sb * p.seq_len;
      seq_t[r] = a_m[r] - seq_base[r];
      seq_n[r] = (p.row_len && a_m[r] < p.M) ? min(p.row_len[sb], p.seq_len) : p.seq_len;
    }
  }

  // one DMA instruction: piece `pi` (0 .. PPW-1) of stage `st`: [A row blocks: hi, lo] then [B row blocks: hi, lo]
  auto issue_piece = [&](int st, int pi) {
    char* sbase = smv2 + (st % NSTAGE) * STAGE_BYTES;
    if (pi < 2 * ARB) {
      const int r = pi >> 1, lo = pi & 1;
      const int blk = wave * ARB + r;
      size_t o;
      bool ok = a_m[r] < p.M;
      if (p.taps > 1) {
        const int k0 = st * 16;
        const int tap = k0 / p.kc, ch = (k0 - tap * p.kc) >> 4;
        int t = seq_t[r] + tap * p.dil - p.pad_left;
        if (p.pad_mode == 1) { t = t < 0 ? -t : t; t = t >= seq_n[r] ? 2 * (seq_n[r] - 1) - t : t; }
        ok = ok && t >= 0 && t < seq_n[r];
        o = ((size_t)ch * q.a_rows + (seq_base[r] + (ok ? t : 0))) * 16 + dunit * 8;
      } else {
        o = ((size_t)st * q.a_rows + min(a_m[r], q.a_rows - 1)) * 16 + dunit * 8;
      }
      const __bf16* src = ok ? (lo ? q.a_lo : q.a_hi) + o : q.zeros;
      GLDS16(src, sbase + lo * A_PLANE + blk * 1024);
    } else {
      const int pb = pi - 2 * ARB;
      const int r = pb >> 1, lo = pb & 1;
      const int blk = wave * BRB + r;
      const size_t o = ((size_t)st * q.npad + (size_t)bn * BN + 32 * blk + lrow) * 16 + dunit * 8;
      GLDS16((lo ? q.b_lo : q.b_hi) + o, sbase + 2 * A_PLANE + lo * B_PLANE + blk * 1024);
    }
  };

  const int ns = q.nstages;
  // fragment read offsets (bytes inside a plane image): row * 32 + (h ^ (row >> 3 & 1)) * 16, row = tile row of lane j
  const int sw = (h ^ ((j >> 3) & 1)) * 16;
  const int a_off = (wm * 64 + j) * 32 + sw;          // + t * 1024 for the second 32-row tile
  const int b_offr = (wn * 128 + j) * 32 + sw;        // + t * 1024 per 32-column tile

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < ns) {
#pragma unroll
      for (int pi = 0; pi < PPW; ++pi) issue_piece(s, pi);
    }

  struct Frag { bf16x8 ah[2], al[2], bh[4], bl[4]; };
  // stage s has landed for the whole workgroup: this wave's pieces by the counted wait (younger stages may stay in
  // flight), everyone's by the barrier; the lgkmcnt(0) retires this wave's fragment reads of stage s - 1, so after the
  // barrier that ring slot may be overwritten
  auto wait_stage = [&](int s) {
    const int issued = min(ns - 1, max(NSTAGE - 1, s + NSTAGE - 2));
    const int pending = issued - s;
    if (pending >= 3) wait_vm_lgkm0<3 * PPW>();
    else if (pending == 2) wait_vm_lgkm0<2 * PPW>();
    else if (pending == 1) wait_vm_lgkm0<PPW>();
    else wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_frags = [&](Frag& f, int s) {
    const char* st = smv2 + (s % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f.ah[t] = *reinterpret_cast<const bf16x8*>(st + a_off + t * 1024);
      f.al[t] = *reinterpret_cast<const bf16x8*>(st + A_PLANE + a_off + t * 1024);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f.bh[t] = *reinterpret_cast<const bf16x8*>(st + 2 * A_PLANE + b_offr + t * 1024);
      f.bl[t] = *reinterpret_cast<const bf16x8*>(st + 2 * A_PLANE + B_PLANE + b_offr + t * 1024);
    }
  };
  // 24 MFMAs of stage i; the DMA of stage i + NSTAGE (ring slot of stage i, free since the last barrier) is issued
  // piecewise in their shadow (an LDS-DMA instruction costs ~100-180 issue cycles on its own, little behind an MFMA)
  auto compute = [&](const Frag& f, int i) {
    const bool more = i + NSTAGE < ns;
    const int nst = i + NSTAGE;
    constexpr int PQ = (PPW + 3) / 4;          // pieces issued after each of the four MFMA groups
    auto dma = [&](int g) {
#pragma unroll
      for (int u = 0; u < PQ; ++u)
        if (g * PQ + u < PPW) issue_piece(nst, g * PQ + u);
    };
#define V2_MFMA3(mt, nt)                                                                                         \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);             \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bl[nt], acc[mt][nt], 0, 0, 0);             \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
    V2_MFMA3(0, 0) V2_MFMA3(0, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma(0);
    __builtin_amdgcn_sched_barrier(0);
    V2_MFMA3(0, 2) V2_MFMA3(0, 3)
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma(1);
    __builtin_amdgcn_sched_barrier(0);
    V2_MFMA3(1, 0) V2_MFMA3(1, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma(2);
    __builtin_amdgcn_sched_barrier(0);
    V2_MFMA3(1, 2) V2_MFMA3(1, 3)
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma(3);
    __builtin_amdgcn_sched_barrier(0);
#undef V2_MFMA3
  };

  // two fragment register sets: the reads of stage i + 1 are in flight while stage i is multiplied
  Frag f0, f1;
  wait_stage(0);
  load_frags(f0, 0);
  for (int i = 0; i < ns; i += 2) {
    if (i + 1 < ns) { wait_stage(i + 1); load_frags(f1, i + 1); }
    compute(f0, i);
    if (i + 1 < ns) {
      if (i + 2 < ns) { wait_stage(i + 2); load_frags(f0, i + 2); }
      compute(f1, i + 1);
    }
  }
  __syncthreads();
  gemm_epilogue_lds<4, WNW, 2, 4>(p, acc, reinterpret_cast<float*>(smv2), bm * 256, bn * BN, wm, wn, wave, lane);
}

// one output tile per workgroup; XCD x owns the row tiles == x (mod 8) (gemm.hip explains the two walk orders)
template <int WNW, int NSTAGE>
__global__ __launch_bounds__(256 * WNW) void gemm_bf16x3_v2_kernel(const GemmV2P q) {
  const GemmKP& p = q.g;
  const int L = blockIdx.x, xcd = L & 7, qq = L >> 3;
  int bn, bm;
  if (p.n_fast) { const int bml = qq / p.nblocks; bn = qq - bml * p.nblocks; bm = bml * 8 + xcd; }
  else { bn = qq / p.mt8; bm = (qq - bn * p.mt8) * 8 + xcd; }
  if (bm >= p.mtiles) return;
  gemm_v2_tile<WNW, NSTAGE>(q, bm, bn, threadIdx.x);
}

// ---- per-stream scratch for the activation planes (grow-only) ----
struct PlaneScratch { void* ptr = nullptr; size_t bytes = 0; };
static std::map<hipStream_t, PlaneScratch> g_scratch;      // guarded by g_scratch_mu: stages of different batches may run on
static const __bf16* g_zero_page = nullptr;                 // different host threads / streams (bench.py's two-stage pipeline)
static std::mutex g_scratch_mu;


// p: fully prepared by gemm_bf16x3_forward (shapes, epilogue, conv parameters); planes: w.wp16 + offset
int gemm_bf16x3_v2_forward(GemmKP p, const void* wplanes, const LinearWeights& w, const GemmArgs& a, hipStream_t stream, double flops,
                           double bytes) {
  IDX_CHECK(w.K % 16 == 0 && (a.taps <= 1 || (w.K / a.taps) % 16 == 0), "v2 needs K % 16 == 0");
  const int xk = a.taps > 1 ? w.K / a.taps : w.K;       // channels of the activation rows
  const size_t plane = (size_t)(xk / 16) * a.M * 16 * sizeof(__bf16);
  std::unique_lock<std::mutex> lock(g_scratch_mu);
  PlaneScratch& sc = g_scratch[stream];      // (map nodes are stable: the reference outlives the lock; a stream has one user)
  if (!a.x_planes && sc.bytes < 2 * plane) {
    IDX_HIP(hipStreamSynchronize(stream));
    if (sc.ptr) IDX_HIP(hipFree(sc.ptr));
    sc.bytes = 2 * plane + (plane >> 2);
    IDX_HIP(hipMalloc(&sc.ptr, sc.bytes));
  }
  if (!g_zero_page) {
    void* z = nullptr;
    IDX_HIP(hipMalloc(&z, 4096));
    IDX_HIP(hipMemset(z, 0, 4096));
    g_zero_page = static_cast<const __bf16*>(z);
  }
  lock.unlock();
  const __bf16* hi = a.x_planes ? static_cast<const __bf16*>(a.x_planes) : static_cast<const __bf16*>(sc.ptr);
  const __bf16* lo = hi + plane / sizeof(__bf16);
  if (!a.x_planes) {
    ProfScope prof(PROF_ELTWISE, stream, 0.0, 8.0 * a.M * (double)xk);
    hipLaunchKernelGGL(split_planes_kernel, dim3(cdiv(a.M, 64), cdiv(xk, 64)), dim3(256), 0, stream, a.x, a.ldx, a.M, xk, const_cast<__bf16*>(hi), const_cast<__bf16*>(lo));
    IDX_LAUNCH_CHECK();
  }
  GemmV2P q;
  q.g = p;
  if (a.y_planes) {
    const int n_out = (a.act == ACT_SWIGLU || a.act == ACT_GATE) ? w.N / 2 : w.N;
    IDX_CHECK(n_out % 16 == 0 && (a.ldy & 3) == 0, "output planes need N_out % 16 == 0");
    q.g.y_hi = static_cast<__bf16*>(a.y_planes);
    q.g.y_lo = q.g.y_hi + plane_elems(a.M, n_out);
  }
  if (a.rope) {
    IDX_CHECK(a.rope_T > 0 && a.rope_cols % 64 == 0 && a.act == ACT_NONE && (a.ldy & 3) == 0, "fused rotary arguments");
    q.g.rope = a.rope; q.g.rope_T = a.rope_T; q.g.rope_cols = a.rope_cols;
  }
  // tile width 256 columns on one 8-wave workgroup per CU (the 128-column / two-workgroup form measured 1.5x slower on every
  // hot-path shape, profiles/r01_gemm_bench.txt, and is gone)
  constexpr int BN = 256;
  q.g.mtiles = cdiv(a.M, 256);
  q.g.mt8 = cdiv(q.g.mtiles, 8);
  q.g.nblocks = cdiv(w.N, BN);
  q.a_hi = hi; q.a_lo = lo; q.a_rows = a.M;
  q.npad = cdiv(w.N, 256) * 256;
  q.b_hi = static_cast<const __bf16*>(wplanes);
  q.b_lo = q.b_hi + (size_t)(w.K / 16) * q.npad * 16;
  q.zeros = g_zero_page;
  q.nstages = w.K / 16;
  const int64_t grid = (int64_t)8 * q.g.nblocks * q.g.mt8;
  IDX_CHECK(grid < (1ll << 31), "grid size");
  ProfScope prof(PROF_GEMM_BF16X3_256x256, stream, flops, bytes);
  constexpr int lds = 4 * (2 * 256 * 32 + 2 * 256 * 32);
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_v2_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_v2_kernel<2, 4>), dim3((unsigned)grid), dim3(512), lds, stream, q);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
