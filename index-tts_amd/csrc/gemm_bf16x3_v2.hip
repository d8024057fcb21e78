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
#include <utility>

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

#ifdef V2_TIMING
// Diagnostic build (tools/gemm_stamps.py; never the shipped library): s_memtime stamps of one wave per workgroup, written to a
// buffer of their own: [workgroup][8] = start, first stage landed, end of the main loop, end of the epilogue (stores drained),
// s_memrealtime x2, stores issued (wave 1 of the workgroup).
static unsigned long long* g_v2_stamps = nullptr;
extern "C" int idxtts_dbg_v2_stamps(void* buf) { g_v2_stamps = static_cast<unsigned long long*>(buf); return 0; }
#define V2_STAMP(k) do { if (q.stamps && tid == 64) q.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define V2_RSTAMP(k) do { if (q.stamps && tid == 64) q.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define V2_STAMP(k)
#define V2_RSTAMP(k)
#endif

struct GemmV2P {
  GemmKP g;                      // shapes, epilogue operands, conv parameters (x unused)
  const __bf16* a_hi; const __bf16* a_lo;   // [K/16][a_rows][16]
  const __bf16* b_hi; const __bf16* b_lo;   // [K/16][npad][16]
  int a_rows, npad, nstages;
  int a_bytes, b_bytes;          // bytes of the hi + lo planes of each operand (buffer descriptors)
  int a_plane, b_plane;          // byte offset of the lo plane
  int cs, nbg;                   // tile walk: column groups over the XCDs (1, 2, 4 or 8) and column blocks per group
#ifdef V2_TIMING
  unsigned long long* stamps;
#endif
};

// Two geometries of the same loop (CFG):
//   1 (the one in use): workgroup = 2 x 2 waves, each 64 x 64 (2 x 2 MFMA tiles): 128 x 128 per workgroup, 256 threads, 16 KiB
//      stages, 3-deep ring (48 KiB), <= 168 registers: THREE workgroups per CU, whose phases (main loop / store-bound epilogue)
//      interleave on a CU; 4x the tiles of the large form, so launches of a few hundred rows still spread over the chip;
//   0: workgroup = 4 x 2 waves, each a 64 x 128 output tile (2 x 4 MFMA tiles): 256 x 256 per workgroup, 512 threads, 32 KiB
//      stages, 4-deep ring, one workgroup per CU (round 2's geometry; measured slower on every shape, see the dispatch).
//   Same stage = 16 k, same four DMA pieces per wave and stage in both.
//
// Round 3: what a stage's instruction stream holds besides its 24 MFMAs decides the kernel (profiles/README.md "Round 3"):
//   * every LDS-DMA is ONE buffer_load_dwordx4 ... lds: the lane's byte offset inside the operand planes is a VGPR computed
//     once per tile (once per tap for the convolution form), the stage offset an SGPR that advances by a constant, the
//     ring slot goes to M0 -- no vector arithmetic per piece (the per-piece 64-bit address arithmetic of the first form was
//     ~20 VALU + ~35 SALU instructions, 100-185 issue cycles each time, four times per stage);
//   * rows outside a convolution's sequence read zeros through the descriptor's range check (offset bit 31 set), rows
//     beyond M re-read row M - 1 (never stored);
//   * the counted waits are the compiler's own s_waitcnt (builtin), not inline asm: with the asm form the compiler could not
//     know that a fragment set had landed and put s_waitcnt lgkmcnt(0) in front of the first MFMA of every stage -- i.e.
//     BEHIND the 12 fragment reads of the NEXT stage it had just issued, which undid the double-buffered fragment registers.
struct V2Rsrc { __amdgpu_buffer_rsrc_t a, b; };

#define V2_WAITCNT(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | (((vm) >> 4) << 14) | 0x70)   /* vmcnt(vm) lgkmcnt(0) */

template <bool TAPS, int EPI, int CFG>
__device__ __forceinline__ void gemm_v2_tile(const GemmV2P& q, const int bm, const int bn, const int tid) {
  constexpr int NSTAGE = CFG ? 3 : 4;      // ring depth (16-k stages): CFG 2 = the 128 x 128 geometry at three workgroups per CU
  constexpr int WMW = CFG ? 2 : 4, WNW = 2, TN = CFG ? 2 : 4;      // waves along M / N, 32-column MFMA tiles per wave
  constexpr int BM = 64 * WMW, BN = 32 * TN * WNW;                 // 256 x 256 or 128 x 128
  static_assert(BM / 32 == WMW * WNW && BN / 32 == WMW * WNW, "one 32-row block of A and of B per wave and plane");
  constexpr int PLANE = BM * 32;               // bytes of one operand plane of a stage (BM = BN rows x 32 B)
  constexpr int STAGE_BYTES = 4 * PLANE;       // [A hi][A lo][B hi][B lo]
  constexpr int PPW = 4;                       // DMA instructions per wave and stage
  const GemmKP& p = q.g;
  extern __shared__ __attribute__((aligned(1024))) char smv2[];

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- DMA role of this lane: row (lane >> 1) of the wave's 32-row block, LDS unit lane & 1, source unit swizzled ----
  const int lrow = lane >> 1;
  const int dunit = (lane & 1) ^ ((lrow >> 3) & 1);        // block bases are multiples of 32: row bit 3 = lrow bit 3
  const int a_m = bm * BM + 32 * wave + lrow;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(q.a_hi), 0, q.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(q.b_hi), 0, q.b_bytes, 0x00020000);
  const int voffB = (bn * BN + 32 * wave + lrow) * 32 + dunit * 16;
  int voffA = min(a_m, q.a_rows - 1) * 32 + dunit * 16;
  int seq_base = 0, seq_t = 0, seq_n = 0;
  if (TAPS) {
    const int sb = a_m / p.seq_len;
    seq_base = sb * p.seq_len;
    seq_t = a_m - seq_base;
    seq_n = (p.row_len && a_m < p.M) ? min(p.row_len[sb], p.seq_len) : p.seq_len;
  }
  // the lane's A offset for tap `tap` (convolution form): row t = seq_t + tap * dil - pad_left of its own sequence, reflected
  // or zero outside [0, seq_n)
  auto tap_voff = [&](int tap) -> int {
    int t = seq_t + tap * p.dil - p.pad_left;
    if (p.pad_mode == 1) { t = t < 0 ? -t : t; t = t >= seq_n ? 2 * (seq_n - 1) - t : t; }
    const bool ok = a_m < p.M && t >= 0 && t < seq_n;
    return ok ? (seq_base + t) * 32 + dunit * 16 : (int)0x80000000;
  };
  V2_STAMP(0); V2_RSTAMP(4);
  const int ns = q.nstages;
  const int sA = q.a_rows * 32, sB = q.npad * 32;          // bytes between two 16-k chunks of a plane
  const int cpt = TAPS ? (p.kc >> 4) : ns;                 // chunks per tap
  // scalar issue state: the next stage to request, its plane offsets, its chunk inside the tap
  int ist = 0, soA = 0, soB = 0, ich = 0, itap = 0, wslot = 0, rslot = 0;      // (ring slots of the next stage to request / to read)
  if (TAPS) voffA = tap_voff(0);
  char* const lds_w = smv2 + wave * 1024;
  auto issue_piece = [&](int pi) {           // piece pi of stage ist: A hi, A lo, B hi, B lo
    auto dst = (__attribute__((address_space(3))) void*)(lds_w + wslot * STAGE_BYTES + pi * PLANE);
    if (pi == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, dst, 16, voffA, soA, 0, 0);
    else if (pi == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, dst, 16, voffA, soA + q.a_plane, 0, 0);
    else if (pi == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, dst, 16, voffB, soB, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, dst, 16, voffB, soB + q.b_plane, 0, 0);
  };
  auto advance = [&]() {
    ++ist;
    wslot = wslot + 1 == NSTAGE ? 0 : wslot + 1;
    soB += sB;
    soA += sA;
    if (TAPS) {
      if (++ich == cpt) { ich = 0; soA = 0; ++itap; voffA = tap_voff(itap); }
    }
  };

  // fragment read offsets (bytes inside a plane image): row * 32 + (h ^ (row >> 3 & 1)) * 16, row = tile row of lane j
  const int sw = (h ^ ((j >> 3) & 1)) * 16;
  const int a_off = (wm * 64 + j) * 32 + sw;          // + t * 1024 for the second 32-row tile
  const int b_offr = (wn * TN * 32 + j) * 32 + sw;    // + t * 1024 per 32-column tile

  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  for (int s = 0; s < NSTAGE; ++s)
    if (s < ns) {
#pragma unroll
      for (int pi = 0; pi < PPW; ++pi) issue_piece(pi);
      advance();
    }

  struct Frag { bf16x8 ah[2], al[2], bh[TN], bl[TN]; };
  // stage s has landed for the whole workgroup: this wave's pieces by the counted wait (younger stages may stay in
  // flight), everyone's by the barrier; the lgkmcnt(0) retires this wave's fragment reads of stage s - 1, so after the
  // barrier that ring slot may be overwritten.  Stages in flight behind s: min(2, ns - 1 - s) (3 behind stage 0).
  auto wait_stage = [&](int s) {
    const int rem = ns - 1 - s;           // stages behind s; at most NSTAGE - 2 of them have been requested
    if (NSTAGE >= 4 && rem >= 2) V2_WAITCNT(2 * PPW);
    else if (NSTAGE >= 3 && rem >= 1) V2_WAITCNT(PPW);
    else V2_WAITCNT(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_frags = [&](Frag& f) {           // the next stage in order
    const char* st = smv2 + rslot * STAGE_BYTES;
    rslot = rslot + 1 == NSTAGE ? 0 : rslot + 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f.ah[t] = *reinterpret_cast<const bf16x8*>(st + a_off + t * 1024);
      f.al[t] = *reinterpret_cast<const bf16x8*>(st + PLANE + a_off + t * 1024);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      f.bh[t] = *reinterpret_cast<const bf16x8*>(st + 2 * PLANE + b_offr + t * 1024);
      f.bl[t] = *reinterpret_cast<const bf16x8*>(st + 3 * PLANE + b_offr + t * 1024);
    }
  };
  // 24 MFMAs of stage i; the DMA of stage i + NSTAGE (ring slot of stage i, free since the last barrier) is issued
  // piecewise between the four MFMA groups
  auto compute = [&](const Frag& f, const bool more) {
#define V2_MFMA3(mt, nt)                                                                                         \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);             \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bl[nt], acc[mt][nt], 0, 0, 0);             \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
    // four MFMA groups (half a row of tiles each) with one DMA piece behind each
    if constexpr (TN == 4) { V2_MFMA3(0, 0) V2_MFMA3(0, 1) } else { V2_MFMA3(0, 0) }
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_piece(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TN == 4) { V2_MFMA3(0, 2) V2_MFMA3(0, 3) } else { V2_MFMA3(0, 1) }
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_piece(1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TN == 4) { V2_MFMA3(1, 0) V2_MFMA3(1, 1) } else { V2_MFMA3(1, 0) }
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_piece(2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TN == 4) { V2_MFMA3(1, 2) V2_MFMA3(1, 3) } else { V2_MFMA3(1, 1) }
    __builtin_amdgcn_sched_barrier(0);
    if (more) { issue_piece(3); advance(); }
    __builtin_amdgcn_sched_barrier(0);
#undef V2_MFMA3
  };

  // two fragment register sets: the reads of stage i + 1 are in flight while stage i is multiplied
  Frag f0, f1;
  {   // stage 0: up to three younger stages in flight
    const int rem = ns - 1;
    if (NSTAGE >= 4 && rem >= 3) V2_WAITCNT(3 * PPW);
    else if (NSTAGE >= 3 && rem >= 2) V2_WAITCNT(2 * PPW);
    else if (rem >= 1) V2_WAITCNT(PPW);
    else V2_WAITCNT(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  V2_STAMP(1);
  load_frags(f0);
  int i = 0;
  // steady state, straight-line: two younger stages in flight behind every wait, a stage to request beside every compute
  // (a branch-free body also keeps the compiler's counter model exact: at a join of paths with different numbers of LDS reads
  // pending it falls back to s_waitcnt lgkmcnt(0) in front of the first MFMA, i.e. behind the next stage's fragment reads)
  for (; i + NSTAGE + 1 < ns; i += 2) {
    V2_WAITCNT((NSTAGE - 2) * PPW);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f1);
    compute(f0, true);
    V2_WAITCNT((NSTAGE - 2) * PPW);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f0);
    compute(f1, true);
  }
  for (; i < ns; i += 2) {     // the last stages: the ring drains
    if (i + 1 < ns) { wait_stage(i + 1); load_frags(f1); }
    else V2_WAITCNT(0);
    compute(f0, ist < ns);
    if (i + 1 < ns) {
      if (i + 2 < ns) { wait_stage(i + 2); load_frags(f0); }
      else V2_WAITCNT(0);
      compute(f1, ist < ns);
    }
  }
  // (the last wait_stage left no DMA in flight; the epilogue synchronises the workgroup itself before it reuses the ring)
  V2_STAMP(2);
  // 32 x 32 C/D layout: register r = (2 hh + qq) * 4 + e of tile (mt, nt) is row mt * 32 + 16 hh + 8 qq + 4 h + e, column nt * 32 + j
  auto write_pass = [&](int ps, float* lds, int RS) {
    const int mt = ps >> 1, hh = ps & 1;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(8 * qq + 4 * h + e) * RS + nt * 32 + j] = acc[mt][nt][(2 * hh + qq) * 4 + e];
  };
  gemm_epilogue_wave<EPI, TN, 3>(p, write_pass, reinterpret_cast<float*>(smv2), bm * BM + wm * 64, bn * BN + wn * TN * 32, wave, lane);
  V2_STAMP(6);
#ifdef V2_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  V2_STAMP(3); V2_RSTAMP(5);
}

// one output tile per workgroup; XCD x owns the row tiles == x (mod 8) (gemm.hip explains the two walk orders)
template <bool TAPS, int EPI, int CFG>
__global__ __launch_bounds__(CFG ? 256 : 512, CFG ? 3 : 2) void gemm_bf16x3_v2_kernel(const GemmV2P q) {
  const GemmKP& p = q.g;
  const int L = blockIdx.x, xcd = L & 7, qq = L >> 3;
  int bn, bm;
  if (p.n_fast) {
    // XCD x = (row class x / cs, column group x % cs): it owns the row tiles == its class (mod 8 / cs) and a contiguous 1 / cs of the
    // column blocks, and walks its columns fastest.  cs = 1: every XCD sweeps all columns of its rows (A read once, the whole B through
    // its L2); cs > 1 where B would not stay in a 4 MiB L2 next to the streaming A: the XCD's B slice stays resident, A is read cs times.
    const int cs = q.cs, cg = xcd % cs, rc = xcd / cs;
    const int bml = qq / q.nbg, bnl = qq - bml * q.nbg;
    bm = bml * (8 / cs) + rc;
    bn = cg * q.nbg + bnl;
    if (bn >= p.nblocks) return;
  } else { bn = qq / p.mt8; bm = (qq - bn * p.mt8) * 8 + xcd; }
  if (bm >= p.mtiles) return;
  gemm_v2_tile<TAPS, EPI, CFG>(q, bm, bn, threadIdx.x);
}

// ---- per-stream scratch for the activation planes (grow-only) ----
struct PlaneScratch { void* ptr = nullptr; size_t bytes = 0; };
// keyed by (device, stream): the null stream exists on every device.  Guarded by g_scratch_mu (stages of different batches run on
// different streams from different host threads).
static std::map<std::pair<int, hipStream_t>, PlaneScratch> g_scratch;
static std::mutex g_scratch_mu;


// idxtts_release_stream: drop what the library keeps for a stream that is going away
int gemm_release_stream_scratch(hipStream_t stream) {
  int dev_id = 0;
  IDX_HIP(hipGetDevice(&dev_id));
  PlaneScratch sc;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    auto it = g_scratch.find(std::make_pair(dev_id, stream));
    if (it == g_scratch.end()) return 0;
    sc = it->second;
    g_scratch.erase(it);
  }
  if (sc.ptr) IDX_HIP(hipFreeAsync(sc.ptr, stream));
  return 0;
}

// p: fully prepared by gemm_bf16x3_forward (shapes, epilogue, conv parameters); planes: w.wp16 + offset
int gemm_bf16x3_v2_forward(GemmKP p, const void* wplanes, const LinearWeights& w, const GemmArgs& a, hipStream_t stream, double flops,
                           double bytes) {
  IDX_CHECK(w.K % 16 == 0 && (a.taps <= 1 || (w.K / a.taps) % 16 == 0), "v2 needs K % 16 == 0");
  const int xk = a.taps > 1 ? w.K / a.taps : w.K;       // channels of the activation rows
  const size_t plane = (size_t)(xk / 16) * a.M * 16 * sizeof(__bf16);
  int dev_id = 0;
  IDX_HIP(hipGetDevice(&dev_id));
  PlaneScratch* scp = nullptr;
  {      // the lock covers the map only (nodes are stable and a stream has one user): growing one stream's scratch stalls nobody else
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    scp = &g_scratch[std::make_pair(dev_id, stream)];
  }
  PlaneScratch& sc = *scp;
  if (!a.x_planes && sc.bytes < 2 * plane) {
    // stream-ordered: the old block is released behind the launches that still read it, without the device-wide synchronisation of
    // hipFree (which would stall every decode lane and acoustic worker of a serving loop)
    if (sc.ptr) IDX_HIP(hipFreeAsync(sc.ptr, stream));
    sc.ptr = nullptr; sc.bytes = 0;
    const size_t nb = 2 * plane + (plane >> 2);
    IDX_HIP(hipMallocAsync(&sc.ptr, nb, stream));
    sc.bytes = nb;
  }
  const __bf16* hi = a.x_planes ? static_cast<const __bf16*>(a.x_planes) : static_cast<const __bf16*>(sc.ptr);
  const __bf16* lo = hi + plane / sizeof(__bf16);
  if (!a.x_planes) {
    static const int cat_split = prof_register("split_planes_kernel");
    ProfScope prof(cat_split, stream, 0.0, 8.0 * a.M * (double)xk);
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
  // geometry (tools/gemm_ab.py, one process, interleaved; profiles/README.md "Round 3"): 128 x 128 tiles, three workgroups per CU,
  // beat the 256 x 256 form on every shape measured -- M = 50 208: N = 512, K = 512 99 vs 117 us, N = 1536 224 vs 252, N = 3072 (SwiGLU)
  // 410 vs 456; M = 10 848, N = 5120, K = 1280 364 vs 435; 8192^2 x 1024 334 vs 347; under-filled grids (M = 750 .. 2066) 1.8-2.5x --
  // three co-resident workgroups overlap one tile's store-bound epilogue with the others' main loops, and four times as many tiles
  // quantise better over 256 CUs.  The 256 x 256 form stays in the source (CFG 0) as the measured alternative; nothing selects it.
  const int cfg = 1;
  const int BMh = cfg ? 128 : 256, BNh = cfg ? 128 : 256;
  q.g.mtiles = cdiv(a.M, BMh);
  q.g.mt8 = cdiv(q.g.mtiles, 8);
  q.g.nblocks = cdiv(w.N, BNh);
  q.a_hi = hi; q.a_lo = lo; q.a_rows = a.M;
  q.npad = cdiv(w.N, 256) * 256;
  q.b_hi = static_cast<const __bf16*>(wplanes);
  q.b_lo = q.b_hi + (size_t)(w.K / 16) * q.npad * 16;
  q.nstages = w.K / 16;
#ifdef V2_TIMING
  q.stamps = g_v2_stamps;
#endif
  const size_t b_plane = (size_t)(w.K / 16) * q.npad * 32;
  IDX_CHECK(2 * plane < (1ull << 31) && 2 * b_plane < (1ull << 31), "operand planes beyond the 2 GiB a buffer offset addresses");
  q.a_plane = (int)plane; q.a_bytes = (int)(2 * plane);
  q.b_plane = (int)b_plane; q.b_bytes = (int)(2 * b_plane);
  // Column groups: the smallest power of two that brings an XCD's B slice (hi + lo planes) under ~3.2 MB, i.e. next to the streaming A
  // in its 4 MiB L2.  Time does not depend on the threshold between 1.5 and 7 MB (s2mel 0.497-0.503 s sequential, 183-185 audio-s/s
  // pipelined); the bytes past the L2 do (profiles/r04_gemm_cs_pmc.txt: mean FETCH_SIZE per launch 289 MiB at 1.5 MB -- A re-read once per
  // group --, 234 at 3.2, 309 at 7 -- the SwiGLU B no longer fits): 3.2 MB.  A B of up to 8 slices walks the same way (its column groups
  // keep their slices resident; walking column blocks slowest re-read A once per 128 columns).
  q.cs = 1;
  static const double cs_bytes = [] { const char* e = getenv("IDXTTS_GEMM_CS_BYTES"); return e ? atof(e) : 3.2e6; }();      // (measurement hook)
  if (!q.g.n_fast && (double)w.N * w.K * 4.0 <= 8.0 * cs_bytes && a.M > w.N) q.g.n_fast = 1;
  if (q.g.n_fast) {
    while (q.cs < 8 && (double)w.N * w.K * 4.0 / q.cs > cs_bytes && q.g.nblocks >= 2 * q.cs) q.cs *= 2;
  }
  q.nbg = cdiv(q.g.nblocks, q.cs);
  const int64_t grid = q.g.n_fast ? (int64_t)8 * q.nbg * cdiv(q.g.mtiles, 8 / q.cs) : (int64_t)8 * q.g.nblocks * q.g.mt8;
  IDX_CHECK(grid < (1ll << 31), "grid size");
  static const int cat = prof_register("gemm_bf16x3_v2_kernel");
  ProfScope prof(cat, stream, flops, bytes);
  const int lds = cfg ? 3 * (4 * 128 * 32) : 4 * (4 * 256 * 32);
  const bool paired = a.act == ACT_SWIGLU || a.act == ACT_GATE;
  IDX_CHECK(!(a.rope && (a.res || paired)), "the rotary epilogue takes no residual and no paired activation");
  IDX_CHECK(!a.row_len || a.seq_len >= 16, "row masks need seq_len >= 16");
  const int epi = a.rope ? EPI_ROPE : paired ? EPI_PAIRED : a.act != ACT_NONE ? EPI_ACT : EPI_PLAIN;
  typedef void (*KernelFn)(const GemmV2P);
#define V2_ROW(T, C) {gemm_bf16x3_v2_kernel<T, EPI_PLAIN, C>, gemm_bf16x3_v2_kernel<T, EPI_ROPE, C>, gemm_bf16x3_v2_kernel<T, EPI_PAIRED, C>, gemm_bf16x3_v2_kernel<T, EPI_ACT, C>}
  static const KernelFn kernels[2][2][4] = {{V2_ROW(false, 0), V2_ROW(true, 0)}, {V2_ROW(false, 1), V2_ROW(true, 1)}};
#undef V2_ROW
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [&] {
    for (int c = 0; c < 2; ++c)
      for (int t = 0; t < 2; ++t)
        for (int e = 0; e < 4; ++e) {
          const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[c][t][e]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   c ? 3 * (4 * 128 * 32) : 4 * (4 * 256 * 32));
          if (r != hipSuccess) attr_err = r;
        }
  });
  IDX_HIP(attr_err);
  hipLaunchKernelGGL(kernels[cfg][a.taps > 1 ? 1 : 0][epi], dim3((unsigned)grid), dim3(cfg ? 256 : 512), lds, stream, q);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
