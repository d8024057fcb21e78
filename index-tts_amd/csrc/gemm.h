// Token-major linear layers on the fp32 matrix core:  Y[M][N] = epi(X[M][K] * W^T + bias).
#pragma once
#include "common.h"

namespace idxtts {

enum GemmAct { ACT_NONE = 0, ACT_GELU_NEW = 1, ACT_SILU = 2, ACT_SWIGLU = 3, ACT_MISH = 4, ACT_GATE = 5 /* tanh(a)*sigmoid(b), packed like SWIGLU */,
               ACT_GELU_ERF = 6 /* nn.GELU(): exact fp32 GEMM only */, ACT_RELU = 7 /* exact fp32 GEMM only */ };

struct LinearWeights {      // device-resident, packed for the MFMA B operand
  const float* wp = nullptr;   // [ceil(N/32)][ceil(K/16)][g2][h2][j32][4]
  const float* bias = nullptr; // [N] (for SWIGLU: [N] in packed row order) or null
  int N = 0, K = 0;
  const void* wp16 = nullptr;  // optional split-bf16 pack [N/128][K/32][hl][128][40] bf16 (gemm_bf16x3.hip)
};

static inline size_t linear_packed_floats(int N, int K) { return (size_t)cdiv(N, 32) * cdiv(K, 16) * 512; }

// w: [N][K] row-major (torch nn.Linear layout).  Same sub-tile format as the conv weights with one tap.
void pack_linear(float* dst, const float* w, int N, int K);
// HF Conv1D layout [K][N] (y = x @ W + b).
void pack_linear_kn(float* dst, const float* w_kn, int K, int N);

struct GemmArgs {
  const float* x = nullptr; int ldx = 0;      // [M][K], row stride ldx floats
  float* y = nullptr; int ldy = 0;            // [M][N] (SWIGLU: [M][N/2])
  // split-bf16 planes (plane_index, common.h; lo plane at + plane_elems(M, channels)), rows = M:
  //   x_planes: the input already split (by its producer) -- x may then be null; only the LDS-DMA kernel takes it
  //   y_planes: the epilogue also writes its output as planes for the next GEMM; y may then be null
  const void* x_planes = nullptr;
  void* y_planes = nullptr;
  // optional rotary embedding fused into the epilogue (LDS-DMA kernel only: check gemm_uses_planes): output columns
  // < rope_cols are (even, odd) pairs rotated by rope[(m % rope_T)][(col / 2) % 32] = (cos, sin)
  const float* rope = nullptr; int rope_T = 0, rope_cols = 0;
  const float* res = nullptr; int ldr = 0;    // optional residual added after the activation
  // optional per-row-group modulation of the OUTPUT (adaLN etc. are handled by the norm kernels, not here)
  int M = 0;
  int act = ACT_NONE;
  float out_scale = 1.0f;
  // token-major Conv1d over sequences of seq_len rows: y[b,t] = sum_tap W[:, tap*Kc + ci] x[b, t + tap*dil - pad_left, ci]
  // (w.K = taps * Kc, Kc % 32 == 0).  pad_mode: 0 zero, 1 reflect (SConv1d, encodec.py:212-228).
  int taps = 1, seq_len = 0, dil = 1, pad_left = 0, pad_mode = 0;
  // optional output row mask: rows with (m % seq_len) >= row_len[m / seq_len] are written as 0 (x * x_mask)
  const int* row_len = nullptr;
  // exact-fp32 kernel only: split the K loop over `ksplit` workgroups per output tile (skinny GEMMs with a very long K:
  // the conformer's input projection, K = 261 632).  y then receives ksplit raw partial slabs [ksplit][M][ldy] -- no bias,
  // activation or residual -- which the caller sums in a fixed order (rows_norm_forward partials), so results stay
  // bitwise reproducible.
  int ksplit = 1;
};

int gemm_tn_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream);      // exact fp32 MFMA
int gemm_bf16x3_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream);  // split-bf16 (3 bf16 MFMAs / product)
size_t linear_bf16x3_packed_bytes(int N, int K);
void pack_linear_bf16x3(void* dst, const float* w, int N, int K);

// Arithmetic of the GEMM-shaped (compute-bound) passes: GEMM_F32 = exact fp32 MFMA everywhere; GEMM_BF16X3 = split-bf16
// for launches with M >= 256 whose weights carry a bf16 pack (s2mel, GPT latent pass).  Decode GEMVs are always fp32.
enum GemmMode { GEMM_F32 = 0, GEMM_BF16X3 = 1 };
void set_gemm_mode(int mode);
int get_gemm_mode();
// dispatches on the mode above
int gemm_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream);
// true when gemm_forward would run this shape on the LDS-DMA split-bf16 kernel (the only consumer / producer of planes)
bool gemm_uses_planes(const LinearWeights& w, const GemmArgs& a);

int gemm_release_stream_scratch(hipStream_t stream);      // idxtts_release_stream: the split-plane scratch kept per stream
int gemm_tn_release_stream_scratch(hipStream_t stream);   // ... and the exact kernel's K-group combine scratch

}  // namespace idxtts
