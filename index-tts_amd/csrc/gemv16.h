#pragma once
#include <algorithm>

#include "common.h"

namespace idxtts {

struct Gemv16Weights {      // packed [ceil(N/16)][ceil(K/16)][64 lanes][4]
  const float* wp = nullptr;
  int N = 0, K = 0;
};

static inline size_t gemv16_packed_floats(int N, int K) { return (size_t)cdiv(N, 16) * cdiv(K, 16) * 256; }
void pack_gemv16_kn(float* dst, const float* w_kn, int K, int N);   // HF Conv1D [K][N]
void pack_gemv16_nk(float* dst, const float* w_nk, int N, int K);   // nn.Linear [N][K]
int gemv16_plan_ksplit(int N, int K);

struct Gemv16Args {
  const float* x = nullptr; int ldx = 0;          // [rows][K], or
  const float* xpart = nullptr; int xparts = 0; int xpart_rows = 0; int ld_xpart = 0;   // x = act(sum_s xpart[s] + xbias)
  const float* xbias = nullptr; int xact = 0;     // 0 none, 1 gelu_new
  float* ypart = nullptr;                         // [ksplit][rows][N]
  int rows = 0;
  int ksplit = 1;
};

int gemv16_forward(const Gemv16Weights& w, const Gemv16Args& a, hipStream_t stream);

}  // namespace idxtts
