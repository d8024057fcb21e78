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
};


__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_GELU_NEW) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.0f + tanhf(u));
  }
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
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
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = act_apply(acc[mt][nt][r] + bias, p.act) * p.out_scale;
        if (p.res) v += p.res[(size_t)m * p.ldr + n];
        if (row_masked(m)) v = 0.0f;
        p.y[(size_t)m * p.ldy + n] = v;
      }
  }
}

// 128x128 workgroup tile, 2x2 waves of 64x64
__device__ __forceinline__ void gemm_epilogue(const GemmKP& p, f32x16 (&acc)[2][2], int bm, int bn, int wm, int wn, int h, int j) {
  gemm_epilogue_t<2, 2>(p, acc, bm * 128 + wm * 64, bn * 128 + wn * 64, h, j);
}

}  // namespace idxtts
