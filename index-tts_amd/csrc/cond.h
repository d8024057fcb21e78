#pragma once
#include "../../include/idxtts.h"
#include "cond_ops.h"
#include "ctx.h"
#include "gemm.h"
#include "norm.h"
#include "prof.h"

namespace idxtts {

struct ConformerLayer {
  const float *mha_g, *mha_b, *conv_g, *conv_b, *ff_g, *ff_b, *fin_g, *fin_b;   // norm_mha / norm_conv / norm_ff / norm_final
  LinearWeights qkv, out, pos, pw1, pw2, ff1, ff2;
  const float *bias_u, *bias_v;                    // [H][dk]
  const float *dw_w, *dw_b, *dwn_g, *dwn_b;        // depthwise conv [D][k], its LayerNorm
};

struct PerceiverLayer {
  LinearWeights to_q, to_kv, to_out, ff1, ff2;     // ff2's K is padded to a multiple of 4 (zero columns)
};

// One prompt encoder = ConformerEncoder + PerceiverResampler (UnifiedVoice.get_conditioning / get_emo_conditioning,
// model_v2.py:627-671); the emotion variant also carries emovec_layer and emo_layer (get_emovec, 897-902).
struct CondModel : ModelBase {
  idxtts_cond_config cfg;
  std::string cprefix, pprefix;
  int dk = 0, F2 = 0, ffi = 0, ffi_pad = 0, inner = 0;
  const float *sub_w = nullptr, *sub_b = nullptr;   // Conv2d(1, D, 3, 2) filters [D][9], bias
  LinearWeights embed;                              // Linear(D * F2 -> D), weights pre-scaled by sqrt(D); bias applied by the reduction
  const float* embed_bias = nullptr;                // sqrt(D) * bias
  const float* pe = nullptr; int pe_len = 0;        // sinusoid table [pe_len][D]
  const float *after_g = nullptr, *after_b = nullptr;
  std::vector<ConformerLayer> layers;
  LinearWeights proj_ctx;
  const float* latents = nullptr;                   // [n][dim]
  std::vector<PerceiverLayer> player;
  const float* pnorm_g = nullptr;
  LinearWeights emovec, emo;                        // emotion variant only

  explicit CondModel(const idxtts_cond_config& c);
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  size_t workspace_bytes(int B, int T) const;
  int forward(const float* feats, const int* lens_host, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st);
};

}  // namespace idxtts
