#pragma once
#include "../../include/idxtts.h"
#include "cond_ops.h"
#include "ctx.h"
#include "gemm.h"
#include "norm.h"
#include "prof.h"

namespace idxtts {

struct ConvNeXtLayer {
  const float *dw_w, *dw_b, *ln_g, *ln_b;
  LinearWeights pw1, pw2;          // pw2 carries the layer scale: diag(gamma) W, gamma * b
};

// The semantic codec's `quantize` (RepCodec: VocosBackbone encoder + factorised VQ), the S_ref step of the prompt block.
struct RepCodecModel : ModelBase {
  idxtts_repcodec_config cfg;
  LinearWeights embed;             // Conv1d(hidden -> dim, k7, pad 3) as a 7-tap token-major GEMM
  const float *norm_g = nullptr, *norm_b = nullptr, *fin_g = nullptr, *fin_b = nullptr;
  std::vector<ConvNeXtLayer> layers;
  LinearWeights enc_out, in_proj, out_proj;
  const float *codebook = nullptr, *codebook_n = nullptr;     // [size][dim] raw and L2-normalised rows

  explicit RepCodecModel(const idxtts_repcodec_config& c) : cfg(c) {}
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  size_t workspace_bytes(int B, int T) const;
  int quantize(const float* x, int B, int T, long long* indices, float* s_out, void* ws, size_t ws_bytes, hipStream_t st);
};

}  // namespace idxtts
