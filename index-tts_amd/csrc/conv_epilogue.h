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

// 4 x 4 transpose across the four lanes of a quad and four registers (two DPP exchange stages): on return a[c] holds what lane
// (quad base + c) had in a[lane & 3].
__device__ __forceinline__ void quad_transpose4(float (&a)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  auto xchg = [](float v, int ctrl_is_xor2) {
    const int x = __float_as_int(v);
    return __int_as_float(ctrl_is_xor2 ? __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, true));
  };
  float r01 = xchg(b0 ? a[0] : a[1], 0), r23 = xchg(b0 ? a[2] : a[3], 0);
  if (b0) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
  float r02 = xchg(b1 ? a[0] : a[2], 1), r13 = xchg(b1 ? a[1] : a[3], 1);
  if (b1) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
}

template <int TM, int TN, typename P>
__device__ __forceinline__ void conv_epilogue(const P& p, f32x16 (&acc)[TM][TN], int row_base, int col_base, int b, int T, int h, int j) {
  const int u_log2 = p.ups_log2, u_mask = (1 << u_log2) - 1;
  const int Cout = p.M >> u_log2;
  const size_t Tout = (size_t)T << u_log2;
  const size_t own_len = p.lens ? (size_t)p.lens[b] * p.len_mul_out : Tout;     // this row's valid output samples
  // Wide form (plain convolutions over rows of a multiple of 4 samples, 16-byte aligned tensors): the four accumulator registers of a
  // group are four consecutive ROWS of one column; a quad transpose turns them into four consecutive COLUMNS of one row per lane, so the
  // residual / accumulate reads and the store are 16 bytes per lane -- a quarter of the memory instructions, eight 128-byte runs each.
  const bool wide = u_log2 == 0 && (T & 3) == 0 && T >= 4 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 &&
                    (!p.res || (reinterpret_cast<uintptr_t>(p.res) & 15) == 0);
  if (wide) {
    const int lane_q = j & 3, jb = j & ~3;
    const int tmax = T - 4;      // (T % 4 == 0, T >= 4): a column group past the row's end reads the last one instead; never stored
    constexpr int GB = TN <= 2 ? 4 : 2;      // row groups per batch: 8 (x 2) 16-byte loads in flight per lane
#pragma unroll
    for (int mg = 0; mg < TM * (4 / GB); ++mg) {
      const int mt = mg / (4 / GB), g0 = (mg % (4 / GB)) * GB;
      // every residual / accumulate operand of this batch of rows is requested before its first store (a load behind a store to the
      // same tensor waits for that store): one round trip per batch instead of one per row group.
      // The loads are branch-free (clamped addresses; rows / columns outside the tile are dropped at the store).
      f32x4 rv[GB][TN], av[GB][TN];
      size_t rowoff[GB];
#pragma unroll
      for (int grp = 0; grp < GB; ++grp) {
        const int m = row_base + mt * 32 + lane_q + 8 * (g0 + grp) + 4 * h;          // the row this lane ends up with
        rowoff[grp] = ((size_t)b * Cout + min(m, p.M - 1)) * Tout;
      }
      if (p.res) {
#pragma unroll
        for (int grp = 0; grp < GB; ++grp)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            rv[grp][nt] = *reinterpret_cast<const f32x4*>(p.res + rowoff[grp] + min(col_base + nt * 32 + jb, tmax));
      } else {
#pragma unroll
        for (int grp = 0; grp < GB; ++grp)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) rv[grp][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (p.accum) {
#pragma unroll
        for (int grp = 0; grp < GB; ++grp)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            av[grp][nt] = *reinterpret_cast<const f32x4*>(p.y + rowoff[grp] + min(col_base + nt * 32 + jb, tmax));
      } else {
#pragma unroll
        for (int grp = 0; grp < GB; ++grp)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) av[grp][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int grp = 0; grp < GB; ++grp) {
        const int m = row_base + mt * 32 + lane_q + 8 * (g0 + grp) + 4 * h;
        const bool row_ok = m < p.M;
        const float bias = p.bias ? p.bias[min(m, p.M - 1)] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const int ga = g0 + grp;
          float t4[4] = {acc[mt][nt][4 * ga], acc[mt][nt][4 * ga + 1], acc[mt][nt][4 * ga + 2], acc[mt][nt][4 * ga + 3]};
          quad_transpose4(t4, j);
          const int n0 = col_base + nt * 32 + jb;
          if (!row_ok || n0 >= T) continue;
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float e = (t4[c] + bias + rv[grp][nt][c]) * p.scale;
            if ((size_t)(n0 + c) >= own_len) e = 0.0f;
            v[c] = e + av[grp][nt][c];
          }
          *reinterpret_cast<f32x4*>(p.y + rowoff[grp] + n0) = v;
        }
      }
    }
    return;
  }
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
