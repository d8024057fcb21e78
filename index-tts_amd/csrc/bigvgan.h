#pragma once
#include "../../include/idxtts.h"
#include "ctx.h"

namespace idxtts {

struct AmpBlock {          // AMPBlock1 (bigvgan.py:31-141)
  int kernel = 3;
  int dil[3] = {1, 3, 5};
  ConvWeights convs1[3], convs2[3];
  const float* alpha[6] = {};   // log-scale SnakeBeta params, activations[0..5]
  const float* beta[6] = {};
};

struct BigVGANModel : ModelBase {
  idxtts_bigvgan_config cfg;
  ConvWeights conv_pre;
  std::vector<ConvWeights> ups;
  std::vector<AmpBlock> blocks;
  const float* post_alpha = nullptr;
  const float* post_beta = nullptr;
  const float* conv_post_w = nullptr;   // [C_last][7]
  const float* up_filter = nullptr;     // 12 taps
  const float* down_filter = nullptr;

  explicit BigVGANModel(const idxtts_bigvgan_config& c);
  int stage_channels(int i) const;
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  size_t max_elems(int B, int Tm) const;
  size_t workspace_bytes(int B, int Tm) const;
  int forward(const float* mel, float* wav, int B, int Tm, void* workspace, size_t workspace_bytes, int clamp,
              int stage_idx, float* stage_out, hipStream_t stream, const int* lens = nullptr);
};

}  // namespace idxtts
