// Helpers shared by the small once-per-prompt models (cond.hip, semantic.hip): staged host tensors -> device weights in the
// exact-fp32 MFMA GEMM pack (+ the split-bf16 pack), workspace carving, and the two launches every layer repeats.
#pragma once
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ctx.h"
#include "gemm.h"
#include "norm.h"

namespace idxtts {
namespace {

int need(std::map<std::string, HostTensor>& t, const std::string& key, std::vector<int64_t> shape, HostTensor** out) {
  auto it = t.find(key);
  if (it == t.end()) IDX_FAIL("missing tensor '" + key + "'");
  if (it->second.shape != shape) IDX_FAIL("tensor '" + key + "' has the wrong shape");
  *out = &it->second;
  return 0;
}

int up(DeviceArena& arena, const std::vector<float>& v, const float** out) {
  float* d = nullptr;
  if (arena.upload(v.data(), v.size(), &d)) return 1;
  *out = d;
  return 0;
}

int vec_from(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& key, int n, const float** out) {
  HostTensor* h = nullptr;
  if (need(t, key, {n}, &h)) return 1;
  return up(arena, h->data, out);
}

// [N][K] host rows (+ bias) -> exact-fp32 MFMA pack; K is padded with zero columns to Kpad (a multiple of 4)
int make_linear(DeviceArena& arena, const float* w, const float* bias, int N, int K, int Kpad, LinearWeights* out) {
  std::vector<float> padded;
  if (Kpad != K) {
    padded.assign((size_t)N * Kpad, 0.0f);
    for (int n = 0; n < N; ++n) std::memcpy(&padded[(size_t)n * Kpad], w + (size_t)n * K, K * sizeof(float));
    w = padded.data();
  }
  std::vector<float> packed(linear_packed_floats(N, Kpad));
  pack_linear(packed.data(), w, N, Kpad);
  if (up(arena, packed, &out->wp)) return 1;
  out->N = N; out->K = Kpad;
  if (N >= 96 && Kpad % 16 == 0) {      // split-bf16 copy: launches of >= 256 rows run on the LDS-DMA kernel in GEMM_BF16X3 mode (gemm_forward)
    std::vector<float> p16((linear_bf16x3_packed_bytes(N, Kpad) + 3) / 4);
    pack_linear_bf16x3(p16.data(), w, N, Kpad);
    const float* d16 = nullptr;
    if (up(arena, p16, &d16)) return 1;
    out->wp16 = d16;
  }
  if (bias) {
    std::vector<float> b(bias, bias + N);
    if (up(arena, b, &out->bias)) return 1;
  }
  return 0;
}

int linear_from(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& prefix, int N, int K, bool bias,
                LinearWeights* out, std::vector<int64_t> wshape = {}) {
  HostTensor *w = nullptr, *b = nullptr;
  if (wshape.empty()) wshape = {N, K};
  if (need(t, prefix + ".weight", wshape, &w)) return 1;
  if (bias && need(t, prefix + ".bias", {N}, &b)) return 1;
  return make_linear(arena, w->data.data(), b ? b->data.data() : nullptr, N, K, (K + 3) & ~3, out);
}

int ln_from(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& prefix, int n, const float** g, const float** b) {
  return vec_from(t, arena, prefix + ".weight", n, g) || vec_from(t, arena, prefix + ".bias", n, b);
}

struct Carver {
  char* base; size_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

int lin(const LinearWeights& w, const float* x, int ldx, float* y, int ldy, int M, hipStream_t st, int act = ACT_NONE,
        const float* res = nullptr, int ldr = 0) {
  GemmArgs g;
  g.x = x; g.ldx = ldx; g.y = y; g.ldy = ldy; g.M = M; g.act = act; g.res = res; g.ldr = ldr;
  // exact fp32 MFMA in GEMM_F32 mode; split-bf16 (3 bf16 MFMAs per product, ~2^-16 per product) for M >= 256 in the default mode
  if ((act == ACT_GELU_ERF || act == ACT_RELU)) return gemm_tn_forward(w, g, st);      // (activations only the exact kernel's epilogue has)
  return gemm_forward(w, g, st);
}

// exact fp32 whatever the mode: where an integer result follows (the semantic codec's nearest-code search)
int lin_exact(const LinearWeights& w, const float* x, int ldx, float* y, int ldy, int M, hipStream_t st, int act = ACT_NONE,
              const float* res = nullptr, int ldr = 0) {
  GemmArgs g;
  g.x = x; g.ldx = ldx; g.y = y; g.ldy = ldy; g.M = M; g.act = act; g.res = res; g.ldr = ldr;
  return gemm_tn_forward(w, g, st);
}

int layer_norm(const float* x, float* y, const float* g, const float* b, int M, int d, hipStream_t st) {
  RowsNormArgs n;
  n.x_in = x; n.ld_in = d; n.y = y; n.ld_y = d; n.M = M; n.d = d; n.mode = NORM_LN; n.eps = 1e-5f; n.g1 = g; n.b1 = b;
  return rows_norm_forward(n, st);
}


}  // namespace
}  // namespace idxtts
