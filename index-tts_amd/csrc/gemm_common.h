// Shared by the fp32 and the split-bf16 GEMM kernels: kernel parameter block, activation functions and the
// accumulator epilogue (bias, activation / paired gate, residual, row mask, store) for a 2x2-wave, 2x2-tile
// 128x128 workgroup tile in the 32x32 MFMA C/D layout.
#pragma once
#include "gemm.h"

namespace idxtts {

struct GemmKP {
  const float* x; const float* wp; const float* bias; const float* res; float* y;
  int M, N, K, ldx, ldy, ldr;
  int kc16;        // 16-wide K chunks in the packed weights
  int mtiles, mt8; // 128-row tiles, ceil(mtiles/8)
  int nblocks, n_fast;
  int act;
  float out_scale;
  int taps, kc, seq_len, dil, pad_left, pad_mode;
  const int* row_len;
  __bf16* y_hi; __bf16* y_lo;      // optional split-bf16 planes of the output ([N_out/16][M][16]); y may be null then
  // optional fused rotary embedding (gemm_epilogue_lds only): columns < rope_cols are (even, odd) pairs rotated by
  // rope[(m % rope_T)][(col / 2) % 32] = (cos, sin)   (gpt_fast/model.py apply_rotary_emb on q and k of a fused qkv)
  const float* rope; int rope_T; int rope_cols;
  int ksplit, ksteps_per_split;    // exact-fp32 kernel: K loop split over blockIdx.y (raw partial slabs)
  // exact-fp32 kernel: K is summed in groups of `kg` 32-k steps (a group accumulates from zero, the groups are added in order), whatever
  // the launch geometry.  Few-tile launches put one group per workgroup (blockIdx.y); the LAST workgroup of an output tile to arrive at
  // sk_cnt[tile] adds the groups' partial tiles from sk_slab in group order and runs the epilogue -- the same sum, bit for bit, as the
  // one-workgroup form.  direct_map: tiles dealt to consecutive workgroup ids (= round-robin over the XCDs) instead of the XCD-aware walk.
  int kg, direct_map;
  float* sk_slab; unsigned* sk_cnt; int sk_slab_bytes;
};


// Hardware-rate forms for the split-bf16 GEMM's fused gates (v_exp_f32 + v_rcp_f32, ~1 ulp each: absolute error < 2e-7, far below
// the 2^-16 of the products they follow).  The exact-fp32 kernels keep libm (gemm_epilogue_t, act_apply).
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(2.0f, sigmoid_fast(2.0f * x), -1.0f); }

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_GELU_NEW) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.0f + tanhf(u));
  }
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_MISH) {   // x * tanh(softplus(x)), softplus threshold 20 as torch
    const float sp = v > 20.0f ? v : log1pf(expf(v));
    return v * tanhf(sp);
  }
  return v;
}


// Wave-tile epilogue in the 32x32 MFMA C/D layout: acc[mt][nt][r] is element
//   row = row0 + mt*32 + (r&3) + 8*(r>>2) + 4*h,  packed column = col0 + nt*32 + j
// (row0/col0 = first row / first packed column of the wave's tile).  Bias, activation or paired gate
// (tiles (2q, 2q+1) = [gate | linear]), residual, row mask, store.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_t(const GemmKP& p, f32x16 (&acc)[TM][TN], int row0, int col0, int h, int j) {
  auto row_masked = [&](int m) -> bool {
    if (!p.row_len) return false;
    const int sb = m / p.seq_len;
    return (m - sb * p.seq_len) >= p.row_len[sb];
  };
  if (p.act == ACT_SWIGLU || p.act == ACT_GATE) {
    static_assert(TN % 2 == 0, "paired activations need an even number of column tiles per wave");
#pragma unroll
    for (int q = 0; q < TN / 2; ++q) {
      const int n0 = col0 + q * 64 + j;               // packed column of the gate
      const int no = ((col0 + q * 64) >> 1) + j;      // output column
      const bool ok = (col0 + q * 64) < p.N;          // N % 64 == 0: 64 packed columns are all in or all out
      const float b0 = (p.bias && ok) ? p.bias[n0] : 0.0f, b1 = (p.bias && ok) ? p.bias[n0 + 32] : 0.0f;
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m >= p.M || !ok) continue;
          const float gte = acc[mt][2 * q][r] + b0, lin = acc[mt][2 * q + 1][r] + b1;
          float v = p.act == ACT_SWIGLU ? (gte / (1.0f + expf(-gte))) * lin : tanhf(gte) * (1.0f / (1.0f + expf(-lin)));
          v *= p.out_scale;
          if (p.res) v += p.res[(size_t)m * p.ldr + no];
          if (row_masked(m)) v = 0.0f;
          p.y[(size_t)m * p.ldy + no] = v;
        }
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int n = col0 + nt * 32 + j;
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.0f;
    // the residual may alias the output (in-place x += ...), so the compiler keeps every load behind the store before it: request a
    // batch of 8 residual values first, then finish and store those 8 elements (a thread reads only the elements it writes)
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float rv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = half * 8 + q;
          const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          rv[q] = (p.res && m < p.M) ? p.res[(size_t)m * p.ldr + n] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = half * 8 + q;
          const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m >= p.M) continue;
          float v = act_apply(acc[mt][nt][r] + bias, p.act) * p.out_scale + rv[q];
          if (row_masked(m)) v = 0.0f;
          p.y[(size_t)m * p.ldy + n] = v;
        }
      }
  }
}

// Hardware-rate activations for the split-bf16 epilogue (errors ~1e-7, far below the 2^-16 of the products before them).
__device__ __forceinline__ float act_apply_fast(float v, int act) {
  if (act == ACT_GELU_NEW) return v * sigmoid_fast(1.5957691216057308f * (v + 0.044715f * v * v * v));   // 0.5 v (1 + tanh u) = v sigmoid(2u)
  if (act == ACT_SILU) return v * sigmoid_fast(v);
  return act_apply(v, act);
}

// Epilogue of the LDS-DMA kernel's 256 x 256 tile (8 waves; wave (wm, wn) holds rows wm*64.., packed columns wn*128..), one wave
// at a time: every wave transposes ITS OWN 64 x 128 accumulator tile through a private 16-row LDS image, four passes of
// 16 rows, so that every global access is row-contiguous (a lane owns 4 adjacent columns: 16-byte bias / residual loads and
// stores, 512-byte row segments) and every per-row quantity is uniform over half a wave.  Round 3 (profiles/README.md):
//   * no workgroup barrier after the first one: the earlier form walked four 64-row slabs with two barriers each, and each slab
//     was a dependent round trip (residual load -> add -> store) of ~3.5 us: 26 % of a K = 512 tile;
//   * the residual rows (or the rotary table rows) of THREE passes are requested before the first pass starts -- the fragment
//     registers of the main loop are dead by then -- and the fourth after the first pass; a residual may alias the output
//     (in-place h += ...), and a load issued behind a store waits for that store's acknowledgement (vmcnt counts both);
//   * output planes leave in plane order: a second read of the finished pass (written back in place) hands every lane 8
//     adjacent columns of one row, 16 rows x 32 bytes of one 16-column chunk are one 512-byte run per plane.
// EPI selects the body at compile time (the kernel is instantiated per kind; a run-time switch inside 32 unrolled row groups
// is what made the first form's code tens of thousands of instructions):
//   EPI_PLAIN   bias, out_scale, residual, row mask          EPI_ROPE    bias, rotary pairs (no residual)
//   EPI_PAIRED  SwiGLU / tanh-sigmoid gate of packed [gate 32 | linear 32] column groups, residual, row mask
//   EPI_ACT     any other activation (rolled loops, operands loaded in place: the rare shapes)
// All waves of the workgroup must call it (one __syncthreads inside, behind the operand requests); `lds_base` = the workgroup's
// LDS (>= waves x 16 x (wave columns + 4) floats), no DMA in flight.
enum { EPI_PLAIN = 0, EPI_ROPE = 1, EPI_PAIRED = 2, EPI_ACT = 3 };

// write_pass(ps, lds, RS): stores rows 16 ps .. 16 ps + 15 of the wave's accumulator tile into lds[row * RS + column] -- the one
// place that knows the MFMA shape's C/D layout (32x32 or 16x16 tiles).
template <int EPI, int TN, int PF, class WritePass>      // PF: passes whose row operands are requested ahead (3, or 1 where registers are short)
__device__ __forceinline__ void gemm_epilogue_wave(const GemmKP& p, WritePass&& write_pass, float* lds_base, const int row0, const int col0,
                                                   const int wave, const int lane) {
  constexpr int WC = 32 * TN;                   // packed columns of the wave's tile (128 or 64)
  constexpr bool PAIRED = EPI == EPI_PAIRED;
  constexpr bool PREF = EPI != EPI_ACT;         // residual / rotary rows prefetched into registers
  constexpr int RS = WC + 4;                    // floats per LDS row: + 4 (row r + 1 starts one 16-byte slot further)
  constexpr int LPR = PAIRED ? WC / 8 : WC / 4; // lanes per row
  constexpr int RPI = 64 / LPR;                 // rows per wave-instruction
  constexpr int NIT = 16 / RPI;                 // instructions per 16-row pass
  float* const lds = lds_base + wave * (16 * RS);
  const int sub = lane / LPR, c = lane % LPR;
  int n_out, nb0, nb1 = 0, lcol, ocol;          // output column, bias columns, LDS column read, LDS column of the finished value
  if (PAIRED) {
    const int g = c >> 3, cc = (c & 7) * 4;     // 64-column packed group [gate 32 | linear 32], offset inside its 32 outputs
    nb0 = col0 + g * 64 + cc; nb1 = nb0 + 32;
    n_out = ((col0 + g * 64) >> 1) + cc;
    lcol = g * 64 + cc; ocol = g * 32 + cc;
  } else {
    nb0 = col0 + c * 4; n_out = nb0; lcol = c * 4; ocol = lcol;
  }
  const int n_lim = PAIRED ? (p.N >> 1) : p.N;
  const bool col_ok = nb0 < p.N;
  const bool vec = col_ok && n_out + 3 < n_lim && ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                   (!p.res || (((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (p.bias && col_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (nb0 + e < p.N) b0[e] = p.bias[nb0 + e];
      if (PAIRED && nb1 + e < p.N) b1[e] = p.bias[nb1 + e];
    }
  }
  const bool use_res = EPI != EPI_ROPE && p.res && vec;
  const bool use_rope = EPI == EPI_ROPE && vec && n_out < p.rope_cols;
  auto fetch = [&](int m) -> f32x4 {            // the row operand of this lane: residual, or (cos, sin) rows of the rotary table
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < p.M) {
      if (EPI == EPI_ROPE) { if (use_rope) v = *reinterpret_cast<const f32x4*>(p.rope + ((size_t)(m % p.rope_T) * 32 + ((n_out >> 1) & 31)) * 2); }
      else if (use_res) v = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldr + n_out);
    }
    return v;
  };
  f32x4 pre[4][NIT];
  auto prefetch = [&](int ps) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) pre[ps][it] = fetch(row0 + ps * 16 + it * RPI + sub);
  };
  if (PREF) {
#pragma unroll
    for (int ps = 0; ps < PF; ++ps) prefetch(ps);
  }
  __syncthreads();                              // every wave has read its last fragments: the ring is free
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int nseq = p.row_len ? (p.M + p.seq_len - 1) / p.seq_len : 1;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    write_pass(ps, lds, RS);
    // row mask of the pass: its 16 rows lie in at most two sequences (seq_len >= 16 is checked on the host)
    int t_first = 0, len_a = 0x7fffffff, len_b = 0x7fffffff, seq_len = 0x7fffffff;
    if (p.row_len) {
      const int mb = row0 + ps * 16, sb = min(mb / p.seq_len, nseq - 1);
      t_first = mb - sb * p.seq_len; seq_len = p.seq_len;
      len_a = p.row_len[sb]; len_b = p.row_len[min(sb + 1, nseq - 1)];
    }
    auto body = [&](const int rr, const f32x4 opnd) {
      const int m = row0 + ps * 16 + rr;
      if (m >= p.M || !col_ok) return;
      f32x4 v = *reinterpret_cast<const f32x4*>(lds + rr * RS + lcol) + b0;
      if (PAIRED) {
        const f32x4 lin = *reinterpret_cast<const f32x4*>(lds + rr * RS + lcol + 32) + b1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = p.act == ACT_SWIGLU ? (v[e] * sigmoid_fast(v[e])) * lin[e] : tanh_fast(v[e]) * sigmoid_fast(lin[e]);
      } else if (EPI == EPI_ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply_fast(v[e], p.act);
      }
      if (EPI == EPI_ROPE && use_rope)           // two (even, odd) pairs per lane
        v = f32x4{v[0] * opnd[0] - v[1] * opnd[1], v[1] * opnd[0] + v[0] * opnd[1], v[2] * opnd[2] - v[3] * opnd[3], v[3] * opnd[2] + v[2] * opnd[3]};
      v *= p.out_scale;
      const int t = t_first + rr;
      const bool masked = t < seq_len ? t >= len_a : t - seq_len >= len_b;
      if (vec) {
        if (EPI != EPI_ROPE && use_res) v += opnd;
        if (masked) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.y) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + n_out) = v;      // (non-temporal stores measured equal)
        if (p.y_hi) *reinterpret_cast<f32x4*>(lds + rr * RS + ocol) = v;
      } else {                                   // ragged / unaligned columns: element by element (no planes, no rotary)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n_out + e < n_lim) {
            float o = v[e];
            if (p.res) o += p.res[(size_t)m * p.ldr + n_out + e];
            if (p.y) p.y[(size_t)m * p.ldy + n_out + e] = masked ? 0.0f : o;
          }
      }
    };
    if (PREF) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) body(it * RPI + sub, pre[ps][it]);
    } else {
#pragma unroll 1
      for (int it = 0; it < NIT; ++it) body(it * RPI + sub, fetch(row0 + ps * 16 + it * RPI + sub));
    }
    if (p.y_hi) {
      // plane order: 16-byte unit u of a chunk's 512-byte run = (row u >> 1, columns 8 (u & 1) ..); a chunk's lanes write 128
      // contiguous bytes per instruction
      constexpr int CH = PAIRED ? WC / 32 : WC / 16;     // 16-column output chunks of this wave
      constexpr int LPC = 64 / CH, NK = 32 / LPC;
      const int chunk = lane / LPC;
      const int n0 = (PAIRED ? (col0 >> 1) : col0) + chunk * 16;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int u = k * LPC + (lane % LPC);
        const int rr = u >> 1, hf = u & 1;
        const int m = row0 + ps * 16 + rr;
        if (m >= p.M || n0 >= n_lim) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + rr * RS + chunk * 16 + hf * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + rr * RS + chunk * 16 + hf * 8 + 4);
        bf16x4_t h0, l0, h1, l1;
        split_bf16_x4(v0, h0, l0);
        split_bf16_x4(v1, h1, l1);
        const size_t o = plane_index(m, n0 + hf * 8, p.M);
        *reinterpret_cast<bf16x8_t*>(p.y_hi + o) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8_t*>(p.y_lo + o) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    if (PREF && ps + PF < 4) prefetch(ps + PF);
  }
}

// 128x128 workgroup tile, 2x2 waves of 64x64
__device__ __forceinline__ void gemm_epilogue(const GemmKP& p, f32x16 (&acc)[2][2], int bm, int bn, int wm, int wn, int h, int j) {
  gemm_epilogue_t<2, 2>(p, acc, bm * 128 + wm * 64, bn * 128 + wn * 64, h, j);
}

}  // namespace idxtts
