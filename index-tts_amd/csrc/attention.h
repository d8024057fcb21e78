#pragma once
#include "common.h"

namespace idxtts {

struct AttnArgs {
  const float* q = nullptr; const float* k = nullptr; const float* v = nullptr; float* o = nullptr;
  // element strides (floats): batch stride, token stride; head h starts at +64*h inside a token
  long q_bs = 0, k_bs = 0, v_bs = 0, o_bs = 0;
  int q_ts = 0, k_ts = 0, v_ts = 0, o_ts = 0;
  int B = 0, H = 0, Sq = 0, Sk = 0, head_dim = 64;
  int causal = 0;                 // key <= query (Sq == Sk, aligned at 0)
  const int* kstart = nullptr;    // [B] first valid key (left padding), or null
  const int* kend = nullptr;      // [B] one past the last valid key (right padding / x_lens), or null
  float scale = 0.125f;
  void* o_planes = nullptr;       // split_bf16 only: write the output as split-bf16 planes over rows b*Sq + q, columns head*64 + d (o may be null)
  int split_bf16 = 0;             // 1: split-bf16 products (3 bf16 MFMAs each, ~2^-16 relative) instead of exact-fp32 MFMAs
};

int flash_attn_forward(const AttnArgs& a, hipStream_t stream);

}  // namespace idxtts
