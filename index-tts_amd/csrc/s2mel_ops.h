#pragma once
#include "common.h"

namespace idxtts {

// in-place rotary on the q and k parts of a fused [M][3*H*64] projection (gpt_fast/model.py:348-360)
int rotary_qk(float* qkv, int M, int H, int seq_len, const float* rope /* [>=seq_len][32][2] */, hipStream_t st);
int silu_rows(float* y, const float* x, size_t n, hipStream_t st);

struct CfmPackArgs {     // builds the 2B stacked [cond | null] DiT input rows (flow_matching.py:89-93, diffusion_transformer.py:215-224)
  float* x_in; int ld;           // [2B*T][C + C + D + S]
  const float* x;                // [B][C][T]
  const float* prompt;           // [B][C][Tp_max] (values beyond prompt_len[b] ignored)
  const int* prompt_len;         // [B]
  const float* cond;             // [B*T][D] = cond_projection(mu)
  const float* cond_null;        // [D]      = cond_projection(0) = its bias
  const float* style;            // [B][S]
  int B, T, C, D, S, Tp_max;
  int x_only = 0;                // 1: rewrite the x columns only (prompt, cond and style columns are the same at every Euler step)
};
int cfm_pack(const CfmPackArgs& a, hipStream_t st);

struct CfmEulerArgs {    // x += dt*((1+cfg)*v_cond - cfg*v_null); x[:, :, :Tp] = 0   (flow_matching.py:104-113)
  float* x;                      // [B][C][T]
  const float* v; int ldv;       // conditional half: [B*T][C] DiT output rows
  const float* v_null = nullptr; // null half (default: the B*v_T rows behind the conditional half)
  const int* prompt_len;
  int B, T, C;
  float dt, cfg_rate;
  // v may hold only the frames t >= v_t0 of every sequence (v_T of them): row (n, t) at n * v_T + (t - v_t0); frames t < v_t0 lie
  // inside every prompt (x stays 0 there).  v_T = 0: all T frames (v_t0 = 0).
  int v_t0 = 0, v_T = 0;
};
int cfm_euler(const CfmEulerArgs& a, hipStream_t st);

// GroupNorm(1 group) over the valid (row_len[b] x C) block of each sequence, then Mish; padded rows -> 0
// (length_regulator.py:51-54 nn.GroupNorm(groups=1) + nn.Mish, per-utterance statistics)
int groupnorm1_mish(float* y, const float* x, const float* gamma, const float* beta, const int* row_len, int B, int T, int C,
                    float eps, float* stats /* [B][2] scratch */, hipStream_t st);
// dst[(n, j)][:cols] = src[(n, t0 + j)][:cols], j < Tn: the frames t >= t0 of every sequence, compacted (row strides ld_dst / ld_src)
int gather_tail_rows(float* dst, int ld_dst, const float* src, int ld_src, int cols, int N, int T, int t0, hipStream_t st);
// x[b][c][t < zero_len[b]] = 0 and copy: used to initialise the Euler state
int cfm_init_state(float* x, const float* z, const int* prompt_len, int B, int C, int T, hipStream_t st);

}  // namespace idxtts
