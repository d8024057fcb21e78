#pragma once
#include "../../include/idxtts.h"
#include "ctx.h"
#include "gemm.h"
#include "prof.h"

namespace idxtts {

struct Conv2dW { const float* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 0, stride_h = 1; };   // BatchNorm folded in
struct FcmResBlock { Conv2dW c1, c2, sc; bool has_sc = false; };

struct CamLayer {
  int cin = 0;
  const float *bn1_s = nullptr, *bn1_t = nullptr;     // nonlinear1: y = relu(x * s + t)
  LinearWeights lin1;                                  // linear1 with nonlinear2's BatchNorm folded in (ReLU in the epilogue)
  LinearWeights local;                                 // cam_layer.linear_local as a k-tap token-major GEMM
  const float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;   // context gate: [bn/2][bn], [growth][bn/2]
};
struct CamBlock {
  std::vector<CamLayer> layers;
  int dil = 1, cin = 0, cout = 0;
  const float *tbn_s = nullptr, *tbn_t = nullptr;      // transit: relu(bn(x)) -> linear
  LinearWeights transit;
};

// CAMPPlus speaker encoder (the global style vector of the prompt block): FCM 2-D convolutional head + CAM dense TDNN.
struct CamPPlusModel : ModelBase {
  idxtts_campplus_config cfg;
  Conv2dW conv1, conv2;
  std::vector<FcmResBlock> res;
  LinearWeights tdnn;                                  // Conv1d(320 -> 128, k5, stride 2) + BatchNorm folded, ReLU
  std::vector<CamBlock> blocks;
  const float *out_s = nullptr, *out_t = nullptr;      // out_nonlinear
  LinearWeights dense;                                 // dense.linear with the affine-free BatchNorm folded in
  int fcm_out = 0, final_c = 0, max_c = 0;

  explicit CamPPlusModel(const idxtts_campplus_config& c) : cfg(c) {}
  bool accepts(const std::string& name) const override { return name.rfind("head.", 0) == 0 || name.rfind("xvector.", 0) == 0; }
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  size_t workspace_bytes(int T) const;
  int forward(const float* feat, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st);
};

}  // namespace idxtts
