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

// Workgroup epilogue through LDS: the accumulators of one wave-row (a slab of TM*32 rows x BN columns) are transposed
// through LDS so that every global access of the epilogue is row-contiguous: a lane owns 4 adjacent columns (one 16-byte
// bias load for the whole tile, 16-byte residual loads and stores; a 256-column row is ONE 1-KiB wave store) and every
// per-row quantity (bounds, row mask) is wave-uniform.  The direct form above issues TM*TN*16 scattered dword stores per
// lane from fully unrolled code (tens of thousands of instructions: 40-50 % of a K = 512 GEMM's time on MI355X).
//   WMW x WNW waves, wave (wm, wn) holds rows wm*TM*32.., packed columns wn*TN*32..; lds: >= TM*32 * (BN + 8) floats.
// All waves of the workgroup must call it (it synchronises); LDS must not be in use by anyone (no DMA in flight).
template <int WMW, int WNW, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmKP& p, f32x16 (&acc)[TM][TN], float* lds, int row_base, int col_base,
                                                  int wm, int wn, int wave, int lane) {
  constexpr int BN = WNW * TN * 32, SLAB = TM * 32, NW = WMW * WNW;
  constexpr int RS = BN + 8;                    // +8 floats: the two row groups a wave writes (h = 0/1: rows +4) land on disjoint banks
  const bool paired = p.act == ACT_SWIGLU || p.act == ACT_GATE;
  const int h = lane >> 5, j = lane & 31;
  const bool vec_ok = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
                      (!p.res || (((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
  for (int slab = 0; slab < WMW; ++slab) {
    __syncthreads();
    if (wm == slab) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            lds[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * RS + wn * TN * 32 + nt * 32 + j] = acc[mt][nt][r];
    }
    __syncthreads();
    const int lpr = paired ? BN / 8 : BN / 4;   // lanes per row
    const int rpi = 64 / lpr;                   // rows per wave-instruction
    const int sub = lane / lpr, c = lane - sub * lpr;
    // column operands of this lane are the same for every row
    int n_out, nb0, nb1 = 0;                    // first output column, bias index (and second bias index for pairs)
    const float* src_off;
    if (paired) {
      const int g = c >> 3, cc = (c & 7) * 4;   // 64-column packed group, offset inside its 32 outputs
      nb0 = col_base + g * 64 + cc; nb1 = nb0 + 32;
      n_out = ((col_base + g * 64) >> 1) + cc;
      src_off = lds + g * 64 + cc;
    } else {
      nb0 = col_base + c * 4;
      n_out = nb0;
      src_off = lds + c * 4;
    }
    const int n_lim = paired ? (p.N >> 1) : p.N;
    const bool col_ok = nb0 < p.N;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias && col_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (nb0 + e < p.N) b0[e] = p.bias[nb0 + e];
        if (paired && nb1 + e < p.N) b1[e] = p.bias[nb1 + e];
      }
    }
    // The residual may alias the output (in-place h += ...): a residual load placed next to its row's store stays behind the
    // previous row's store, and its data then waits for that store's acknowledgement (vmcnt counts loads and stores in issue
    // order) -- one HBM write round trip per row.  All residual rows of this wave and slab are requested first instead.
    constexpr int MAXIT = SLAB / NW;
    f32x4 rres[MAXIT];
    const bool res_vec = p.res && vec_ok && col_ok && n_out + 3 < n_lim;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int rr = wave * (SLAB / NW) + sub + it * rpi;
      const int m = row_base + slab * SLAB + rr;
      rres[it] = (res_vec && rr < (wave + 1) * (SLAB / NW) && m < p.M) ? *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldr + n_out)
                                                                       : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int rr = wave * (SLAB / NW) + sub + it * rpi;
      if (rr >= (wave + 1) * (SLAB / NW)) continue;
      const int m = row_base + slab * SLAB + rr;
      if (m >= p.M || !col_ok) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(src_off + rr * RS) + b0;
      if (paired) {
        const f32x4 lin = *reinterpret_cast<const f32x4*>(src_off + rr * RS + 32) + b1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = p.act == ACT_SWIGLU ? (v[e] * sigmoid_fast(v[e])) * lin[e] : tanh_fast(v[e]) * sigmoid_fast(lin[e]);
      } else if (p.act != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
      }
      if (p.rope && n_out < p.rope_cols) {       // two (even, odd) pairs per lane
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rope + ((size_t)(m % p.rope_T) * 32 + ((n_out >> 1) & 31)) * 2);
        v = f32x4{v[0] * cs[0] - v[1] * cs[1], v[1] * cs[0] + v[0] * cs[1], v[2] * cs[2] - v[3] * cs[3], v[3] * cs[2] + v[2] * cs[3]};
      }
      v *= p.out_scale;
      bool masked = false;
      if (p.row_len) {
        const int sb = m / p.seq_len;
        masked = (m - sb * p.seq_len) >= p.row_len[sb];
      }
      float* dst = p.y + (size_t)m * p.ldy + n_out;
      if (vec_ok && n_out + 3 < n_lim) {
        if (p.res) v += rres[it];
        if (masked) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.y) *reinterpret_cast<f32x4*>(dst) = v;
        if (p.y_hi) {       // n_out % 4 == 0: the four columns share a 16-k chunk
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          bf16x4_t hi, lo;
          split_bf16_x4(v, hi, lo);
          const size_t o = plane_index(m, n_out, p.M);
          *reinterpret_cast<bf16x4_t*>(p.y_hi + o) = hi;
          *reinterpret_cast<bf16x4_t*>(p.y_lo + o) = lo;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n_out + e < n_lim) {
            float o = v[e];
            if (p.res) o += p.res[(size_t)m * p.ldr + n_out + e];
            if (p.y) dst[e] = masked ? 0.0f : o;
          }
      }
    }
  }
}

// 128x128 workgroup tile, 2x2 waves of 64x64
__device__ __forceinline__ void gemm_epilogue(const GemmKP& p, f32x16 (&acc)[2][2], int bm, int bn, int wm, int wn, int h, int j) {
  gemm_epilogue_t<2, 2>(p, acc, bm * 128 + wm * 64, bn * 128 + wn * 64, h, j);
}

}  // namespace idxtts
