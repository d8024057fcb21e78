#pragma once
#include "common.h"

namespace idxtts {

enum NormMode { NORM_NONE = 0, NORM_LN = 1, NORM_LN_LN = 2, NORM_ADA_RMS = 3, NORM_MOD_LN = 4 };

struct RowsNormArgs {
  // x = x_in[m] + add_bias + sum_s partials[s][m]
  const float* x_in = nullptr; int ld_in = 0;
  int in_rows_per_batch = 0; long in_batch_stride = 0;   // if >0: row m reads x_in + (m / rpb) * in_batch_stride + (m % rpb) * ld_in
  const float* add_bias = nullptr;
  const float* partials = nullptr; int num_partials = 0; int partial_rows = 0; int ld_partial = 0;
  float* x_out = nullptr; int ld_out = 0;       // optional write-back of x
  float* y = nullptr; int ld_y = 0;             // normalised output
  void* y_planes = nullptr;                     // optional: y also/only as split-bf16 planes (plane_index, common.h; rows = M); y may be null
  int in_frag = 0, y_frag = 0;                  // x_in / y are A-fragment images over d (frag_index, common.h) instead of row-major
  int M = 0, d = 0;
  int mode = NORM_LN;
  float eps = 1e-5f, eps2 = 1e-5f;
  const float* g1 = nullptr; const float* b1 = nullptr;
  const float* g2 = nullptr; const float* b2 = nullptr;
  // per-batch modulation vectors [B][ld_mod]: ADA_RMS: (weight, bias); MOD_LN: (shift, scale)
  const float* mod_a = nullptr; const float* mod_b = nullptr; int ld_mod = 0; int rows_per_batch = 0;
};

int rows_norm_forward(const RowsNormArgs& a, hipStream_t stream);

}  // namespace idxtts
