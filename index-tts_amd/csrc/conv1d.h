// Channels-first Conv1d / ConvTranspose1d as an implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32): M = output channels, N = time, K = (input channel, tap).
#pragma once
#include "common.h"

namespace idxtts {

constexpr int CONV_KC = 16;        // input channels per K-chunk
constexpr int CONV_MT = 32;        // rows per packed weight sub-tile
constexpr int CONV_SUB = 512;      // floats per packed sub-tile: [g2][h2][i32][4]
constexpr int CONV_MAX_HALO = 64;  // (K-1)*dil upper bound supported by the tile loader

enum ConvPadMode { PAD_ZERO = 0, PAD_REFLECT = 1 };

struct ConvWeights {           // packed, device-resident
  const float* wp = nullptr;   // [ceil(M/32)][nchunk][K][g2][h2][i32][4]
  const void* wp16 = nullptr;  // optional split-bf16 images [ceil(M/32)][nchunk][K][hl][g2][i32][8] (conv1d_bf16x3.hip)
  const float* bias = nullptr; // [Cout_real] or null
  int M = 0;                   // GEMM rows (= Cout, or Cout*u for a transposed conv)
  int Cin = 0;
  int K = 0;                   // taps
  int nchunk = 0;              // ceil(Cin/16)
  int ups = 1;                 // u>1: rows are (co*u + r), output written interleaved y[co][s*u + r]
};

static inline size_t conv_packed_floats(int M, int Cin, int K) {
  return (size_t)cdiv(M, CONV_MT) * cdiv(Cin, CONV_KC) * K * CONV_SUB;
}

// Host-side packing (ctx creation time, not on the hot path).
// w: [Cout][Cin][K] (torch Conv1d layout).
void pack_conv1d(float* dst, const float* w, int Cout, int Cin, int K);
// w: [Cin][Cout][Kt] (torch ConvTranspose1d layout), stride u, padding (Kt-u)/2, Kt == 2u.
// Produces the equivalent 3-tap conv with M = Cout*u rows (row co*u+r, taps over x[s-1],x[s],x[s+1]).
void pack_conv_transpose1d(float* dst, const float* w, int Cin, int Cout, int Kt, int u);

struct ConvArgs {
  const float* x = nullptr;    // [B][Cin][T]
  float* y = nullptr;          // [B][Cout][T*ups]
  const float* res = nullptr;  // optional residual, same layout as y
  int B = 0, T = 0;            // T = input length = GEMM columns per batch row
  int dil = 1;
  int pad_left = 0;            // y[t] taps x[t + k*dil - pad_left]
  int pad_mode = PAD_ZERO;
  float scale = 1.0f;          // v = (acc + bias + res) * scale
  int accum = 0;               // y = accum ? y + v : v
  // ragged batches: output samples t >= lens[b] * len_mul_out are written as 0 (a shorter row's tail stays zero, so the next
  // layer's zero padding at that row's own end is what a B = 1 call would see)
  const int* lens = nullptr; int len_mul_out = 1;
};

int conv1d_forward(const ConvWeights& w, const ConvArgs& a, hipStream_t stream);
// split-bf16 path (taken by conv1d_forward when the GEMM mode is GEMM_BF16X3 and w.wp16 is set)
void pack_conv_bf16x3(void* dst, const float* packed_f32, size_t n_subtiles);
int conv1d_bf16x3_forward(const ConvWeights& w, const ConvArgs& a, hipStream_t stream);

}  // namespace idxtts
