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
  // "relative_key" distance embedding (HF Wav2Vec2BertSelfAttention): scores[i][j] += scale * q_i . rel_key[clamp(j - i, -rel_left,
  // rel_right) + rel_left]; rel_key [rel_left + rel_right + 1][64] fp32, one table for every head (<= 96 rows); non-causal self-attention
  // only.  The kernel multiplies the table by its queries once per workgroup (three more 32-row "key" tiles through the same MFMAs) and
  // keeps the products in LDS: a tile of keys outside a wave's band adds one per-query constant, a tile inside it a per-element lookup.
  const float* rel_key = nullptr; int rel_left = 0, rel_right = 0;
};

int flash_attn_forward(const AttnArgs& a, hipStream_t stream);

// The same attention (non-causal, right-padded keys) reading q / k / v as SPLIT-BF16 PLANES of one [Mrows][*] projection output
// (plane_index(row, col, Mrows), common.h: what the LDS-DMA GEMM's epilogue writes): K / V tiles travel HBM -> LDS by
// global_load_lds through a 3-slot ring, two tiles ahead of the MFMAs, with no register staging and no in-loop splitting.
struct AttnPlanesArgs {
  const void* planes = nullptr;   // hi plane; the lo plane follows at + plane_elems(Mrows, ncols)
  int Mrows = 0, ncols = 0;       // rows / columns of the projection output (B * T, 3 * H * 64)
  int q_col = 0, k_col = 0, v_col = 0;   // first column of q / k / v; head h at + 64 h
  int T = 0;                      // rows (= keys) per batch row
  int q_row0 = 0, Sq = 0;         // queries of batch row b: rows b * T + q_row0 + [0, Sq)
  int B = 0, H = 0;
  const int* kend = nullptr;      // [B] valid keys (<= T) or null
  float scale = 0.125f;
  float* o = nullptr; long o_bs = 0; int o_ts = 0;      // fp32 output rows (batch stride, token stride), or null
  void* o_planes = nullptr;       // and / or planes over rows b * Sq + q, columns head * 64 + d
  int nqblk = 0;                  // (set by the launcher) query blocks of 128 per (batch row, head) pair
};
int flash_attn_planes_forward(const AttnPlanesArgs& a, hipStream_t stream);

}  // namespace idxtts
