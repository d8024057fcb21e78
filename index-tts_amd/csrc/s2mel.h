#pragma once
#include "../../include/idxtts.h"
#include "attention.h"
#include "ctx.h"
#include "decode.h"
#include "gemm.h"
#include "norm.h"
#include "prof.h"
#include "s2mel_ops.h"

namespace idxtts {

struct DiTBlock {
  LinearWeights wqkv, wo, w13, w2, skip_a, skip_b;   // skip_a: W[:, :D] (+bias), skip_b: W[:, D:]
  const float *attn_g = nullptr, *ffn_g = nullptr;    // RMSNorm weights
};

struct WNLayer {
  LinearWeights in_gate;   // [2Wh (gate-packed)][k*Wh], bias supplied per Euler step
  LinearWeights res, skip; // 1x1: rows [:Wh] and [Wh:] of res_skip (last layer: skip only)
  bool has_res = true;
};

struct CondBuffers { float *a, *b, *s, *stats; int *idx_code, *idx_row, *idx_interp, *tlen; size_t bytes; };

struct S2MelModel : ModelBase {
  idxtts_s2mel_config cfg;
  std::vector<DiTBlock> blocks;
  const float* final_g = nullptr;
  LinearWeights mod_all, cond_proj, merge, temb0, temb2, t2emb0, t2emb2, skiplin_a, skiplin_b, conv1, res_proj, final_lin,
      final_mod, conv2, wn_cond;
  std::vector<WNLayer> wn;
  // length regulator / gpt_layer / codec
  LinearWeights lr_in, lr_out;
  std::vector<LinearWeights> lr_conv;
  std::vector<const float*> lr_gn_g, lr_gn_b;
  LinearWeights gl[3];
  const float* vq_table = nullptr;    // [codebook_size][codec_hidden]
  const float* rope = nullptr; int rope_len = 0;
  int ffn = 0;

  explicit S2MelModel(const idxtts_s2mel_config& c);
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;

  size_t cfm_workspace_bytes(int B, int T, int n_steps) const;
  int cfm(const float* mu, const int* x_lens_host, const float* prompt, const int* prompt_lens_host, int Tp_max, const float* style,
          const float* z, const float* t_emb, const float* dt_host, int n_steps, float cfg_rate, float* out, int B, int T,
          void* ws, size_t ws_bytes, hipStream_t st);
  int estimator(const float* x, const float* prompt, const int* prompt_lens_host, int Tp_max, const int* x_lens_host, const float* t_emb,
                const float* style, const float* mu, float* out_tm, int B, int T, void* ws, size_t ws_bytes, hipStream_t st);
  size_t cond_workspace_bytes(int B, int M, int Tg) const;
  int regulate_rows(const float* s_rows, const CondBuffers& w, int B, int M, int Tg, float* cond_out, hipStream_t st);
  int regulate(const float* S, const int* in_lens_host, const int* target_lens_host, int B, int M, int Tg, float* cond_out, void* ws,
               size_t ws_bytes, hipStream_t st);
  int prepare_cond(const float* latent, const long long* codes, const int* code_lens_host, const int* target_lens_host, int B, int M,
                   int Tg, float* cond_out, void* ws, size_t ws_bytes, hipStream_t st);
};

// the two CFG halves of the solver on two streams (s2mel.hip, dit_eval_halves); default OFF (include/idxtts.h::idxtts_s2mel_set_overlap)
void set_s2mel_overlap(int on);
int get_s2mel_overlap();
int s2mel_release_stream(hipStream_t st);      // idxtts_release_stream

}  // namespace idxtts
