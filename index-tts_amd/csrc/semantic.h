#pragma once
#include "../../include/idxtts.h"
#include "cond_ops.h"
#include "ctx.h"
#include "gemm.h"
#include "norm.h"
#include "prof.h"

namespace idxtts {

struct W2VLayer {
  const float *ffn1_g, *ffn1_b, *att_g, *att_b, *conv_g, *conv_b, *dwn_g, *dwn_b, *ffn2_g, *ffn2_b, *fin_g, *fin_b;
  LinearWeights ffn1_in, ffn1_out, qkv, out, pw1, pw2, ffn2_in, ffn2_out;
  const float* dist;        // distance_embedding [left + right + 1][head_dim]
  const float* dw_w;        // depthwise conv [D][k]
};

// The semantic feature encoder of the prompt block: the first `num_layers` conformer layers of w2v-bert-2.0
// (HF Wav2Vec2BertModel; hidden_states[num_layers] is what the reference reads, infer_v2.py:381-408), followed by the
// (x - mean) / std normalisation of `get_emb`.
struct W2VBertModel : ModelBase {
  idxtts_w2vbert_config cfg;
  int dk = 0;
  const float *fp_g = nullptr, *fp_b = nullptr;
  LinearWeights proj;
  std::vector<W2VLayer> layers;
  const float *mean = nullptr, *inv_std = nullptr;     // optional: semantic_mean / 1 / semantic_std

  explicit W2VBertModel(const idxtts_w2vbert_config& c) : cfg(c) {}
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  size_t workspace_bytes(int B, int T) const;
  int forward(const float* feats, const int* lens_host, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st);
};

}  // namespace idxtts
