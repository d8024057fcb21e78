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

// Decode GEMV (gemv_fx.hip): activations as A-fragment images (frag_index, common.h), K split across the waves of one
// workgroup, LayerNorm folded into the weights + epilogue, bias / gelu_new / residual epilogue, final output (no slabs).
struct GemvFXArgs {
  const float* xf = nullptr; int rows = 0;        // fragment images [ceil(rows/16)][ceil(K/16)][64][4], padding rows/k zero
  const float* bias = nullptr;                    // [N]; folded layers: c = ln_b . W + bias
  const float* colsum = nullptr; float ln_eps = 1e-5f;   // folded layers: colsum(diag(ln_g) W); weights = diag(ln_g) W
  int act = 0;                                    // 0 none, 1 gelu_new
  const float* res = nullptr;                     // residual in the layout of y (may alias y)
  float* y = nullptr; int y_frag = 0; int ldy = 0;   // y_frag: fragment images over N, else row-major [rows][ldy]
  int ksb = 1;                                    // > 1 (gemv_fx_ksb): y = raw K-slice partial sums [ksb][rows][N]; no epilogue operands
  int dbg = 0;                                    // ablation mask for tools/gemv_probe.hip only
};
int gemv_fx_ksb(int N, int K);
int gemv_fx_combine(const float* slab, int ksb, int rows, int N, const float* bias, const float* res, float* y, hipStream_t stream);
void gemv_fx_plan(int N, int K, int rows, int* ntw, int* kw);
int gemv_fx_forward(const Gemv16Weights& w, const GemvFXArgs& a, hipStream_t stream);

}  // namespace idxtts
