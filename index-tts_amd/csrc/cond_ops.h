// Kernels of the prompt-conditioning encoders (conformer + perceiver) that are not GEMMs or LayerNorms.
#pragma once
#include "common.h"

namespace idxtts {

// Conv2dSubsampling2's conv (subsampling.py:146-149, 176-178): x [B][T][F] -> a [B*T2][C*F2] with T2 = (T-3)/2+1,
// F2 = (F-3)/2+1, a[(b,t2)][c*F2 + f2] = relu(bias[c] + sum_{i,j<3} w[c][i][j] * x[b][2*t2+i][2*f2+j])
// (= conv(x.unsqueeze(1)).transpose(1,2).view(b, t, c*f): the operand of the `out` Linear).
int sub2_conv_relu(float* a, const float* x, const float* w /* [C][3][3] */, const float* bias /* [C] */, int B, int T, int F, int C,
                   hipStream_t st);

// Multi-head attention over short sequences, exact fp32, generic head_dim (<= 128), one wave per (batch, head, query).
//   scores[j] = scale * ( (q + bias_u[h]) . k[j]  +  (q + bias_v[h]) . pos[j] )        (rel-pos form, attention.py:266-312;
//               pos / bias_u / bias_v null -> plain q . k)
//   keys j >= kend[b] are excluded (masked_fill(-inf) before the softmax, 0 after it); o = softmax(scores) v.
struct SeqAttnArgs {
  const float* q = nullptr; int ldq = 0; long q_bs = 0;     // row stride, batch stride (floats); head h at + h*dk
  const float* k = nullptr; int ldk = 0; long k_bs = 0;
  const float* v = nullptr; int ldv = 0; long v_bs = 0;
  const float* pos = nullptr; int ldp = 0;                  // [Sk][H*dk] shared by every batch row, or null
  const float* bias_u = nullptr; const float* bias_v = nullptr;   // [H][dk]
  float* o = nullptr; int ldo = 0; long o_bs = 0;
  const int* kend = nullptr;                                // [B] device, or null
  // "relative_key" distance embedding (HF Wav2Vec2BertSelfAttention, modeling_wav2vec2_bert.py relative_key branch):
  //   scores[j] += scale * q . rel_key[clamp(j - i, -rel_left, rel_right) + rel_left],  rel_key [rel_left+rel_right+1][dk],
  //   one table shared by every head; null = off
  const float* rel_key = nullptr; int rel_left = 0, rel_right = 0;
  int B = 0, H = 0, Sq = 0, Sk = 0, dk = 0;
  float scale = 1.0f;
};
int seq_attn_forward(const SeqAttnArgs& a, hipStream_t st);

// x[m][:] = 0 for rows with (m % T) >= len[m / T]   (masked_fill_(~mask_pad, 0), conformer_encoder.py:131-132)
int mask_rows(float* x, int M, int d, int T, const int* len, hipStream_t st);

// ConvolutionModule after pointwise_conv1 (conformer_encoder.py:147-158): pw [B*T][2D] ->
//   g = pw[:, :D] * sigmoid(pw[:, D:])  (GLU);  depthwise conv k (zero pad (k-1)/2 at the ends of each T-row sequence) + bias;
//   LayerNorm(D, eps 1e-5) * gamma + beta;  SiLU.   y [B*T][D]
// pad_left < 0: (k-1)/2 (ESPnet "same" padding); pad_left = k-1: the causal form of Wav2Vec2BertConvolutionModule (all padding
// on the left).  bdw may be null (depthwise conv without bias).
int glu_dwconv_ln_silu(float* y, float* pw /* consumed: its a half is overwritten by a * sigmoid(gate) */, const float* wdw /* [D][k] */, const float* bdw, const float* gamma, const float* beta,
                       int B, int T, int D, int k, hipStream_t st, int pad_left = -1);

// ConvNeXt block head (kmeans/vocos.py:507-517): depthwise Conv1d k ("same" zero padding) + bias over token-major rows x [B*T][D],
// then LayerNorm(D, eps) * gamma + beta.
int dwconv_ln(float* y, const float* x, const float* wdw /* [D][k] */, const float* bdw, const float* gamma, const float* beta, int B, int T,
              int D, int k, float eps, hipStream_t st);

// ctx[b][i] = lat[b][i] (i < n), ctx[b][n + t] = x[b][t]   (perceiver.py:305-306 cross_attn_include_queries)
int concat_latents_ctx(float* ctx, const float* lat, const float* x, int B, int n, int T, int d, hipStream_t st);

// GEGLU (perceiver.py:174-177): y[m][j] = gelu_erf(in[m][F + j]) * in[m][j], j < F; y[m][F..ldy) = 0
int geglu(float* y, int ldy, const float* in, int M, int F, hipStream_t st);

// perceiver RMSNorm (perceiver.py:150-159): y = x / max(||x||_2, 1e-12) * sqrt(d) * gamma
int l2norm_scale(float* y, const float* x, const float* gamma, int M, int d, hipStream_t st);

// out[b][:] = base[b][:] + alpha * (emo[b][:] - base[b][:])   (merge_emovec, model_v2.py:904-910)
int lerp_rows(float* out, const float* base, const float* emo, float alpha, size_t n, hipStream_t st);

}  // namespace idxtts
