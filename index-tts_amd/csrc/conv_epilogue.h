// Epilogue of the channels-first Conv1d kernels (conv1d.hip, conv1d_bf16x3.hip): bias, residual, scale, ragged-row mask,
// (accumulate), store of a wave's TM x TN accumulator tiles in the 32x32 MFMA C layout (lane j = output column n, register r = row).
//
// The residual (and the accumulate operand) may alias the output, so a load written next to its store could not be moved by the
// compiler: the tile's epilogue used to be a chain of dependent load -> add -> store round trips (measured on the narrow vocoder
// convolutions: 414 of 738 us per launch went to the residual reads alone).  Here every element's operands of
// a quarter of an accumulator tile (4 rows x TN tiles) are requested first and consumed afterwards: a thread still reads exactly the
// elements it writes, before it writes them.
#pragma once
#include "common.h"

namespace idxtts {

template <int TM, int TN, typename P>
__device__ __forceinline__ void conv_epilogue(const P& p, f32x16 (&acc)[TM][TN], int row_base, int col_base, int b, int T, int h, int j) {
  const int u_log2 = p.ups_log2, u_mask = (1 << u_log2) - 1;
  const int Cout = p.M >> u_log2;
  const size_t Tout = (size_t)T << u_log2;
  const size_t own_len = p.lens ? (size_t)p.lens[b] * p.len_mul_out : Tout;     // this row's valid output samples
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {       // 4 accumulator registers (rows) x TN tiles per batch: enough loads in flight, few registers
      float rv[4][TN], av[4][TN];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = grp * 4 + q;
        const int m = row_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const size_t rowoff = ((size_t)b * Cout + (m >> u_log2)) * Tout + (m & u_mask);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const int n = col_base + nt * 32 + j;
          const bool ok = m < p.M && n < T;
          const size_t idx = rowoff + ((size_t)n << u_log2);
          rv[q][nt] = (ok && p.res) ? p.res[idx] : 0.0f;
          av[q][nt] = (ok && p.accum) ? p.y[idx] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = grp * 4 + q;
        const int m = row_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int co = min(m, p.M - 1) >> u_log2, ph = m & u_mask;
        const float bias = p.bias ? p.bias[co] : 0.0f;
        const size_t rowoff = ((size_t)b * Cout + (m >> u_log2)) * Tout + ph;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const int n = col_base + nt * 32 + j;
          if (m >= p.M || n >= T) continue;
          float v = (acc[mt][nt][r] + bias + rv[q][nt]) * p.scale;
          if (((size_t)n << u_log2) + ph >= own_len) v = 0.0f;
          p.y[rowoff + ((size_t)n << u_log2)] = v + av[q][nt];
        }
      }
    }
  }
}

}  // namespace idxtts
