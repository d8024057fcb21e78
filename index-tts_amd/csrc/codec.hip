// The semantic codec's `quantize` on MI355X: S_ref of the prompt block (`_, S_ref = self.semantic_codec.quantize(spk_cond_emb)`,
// infer_v2.py:637).
//
// Reference: RepCodec.quantize                indextts/utils/maskgct/models/codec/kmeans/repcodec_model.py:179-199
//            VocosBackbone / ConvNeXtBlock    indextts/utils/maskgct/models/codec/kmeans/vocos.py:719-782, 468-526
//            ResidualVQ (one quantizer)       indextts/utils/maskgct/models/codec/amphion_codec/quantize/residual_vq.py:68-140
//            FactorizedVectorQuantize         .../quantize/factorized_vector_quantize.py:66-119 (in_project, L2-normalised nearest
//                                             code, z_e + (z_q - z_e), out_project)
//
// Token-major rows [B*T][channels]; the k7 input convolution runs as a 7-tap GEMM over shifted rows (zero padding at both ends of
// every sequence), every linear on the exact-fp32 MFMA GEMM, the nearest-code search one workgroup per frame over the 8192
// normalised codes with the reference's own distance expression.  Runs once per prompt.
#include <cmath>

#include "codec.h"
#include "model_util.h"

namespace idxtts {

bool RepCodecModel::accepts(const std::string& name) const {
  return name.rfind("encoder.", 0) == 0 || name.rfind("quantizer.quantizers.0.", 0) == 0;
}

int RepCodecModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int Hs = cfg.hidden_size, D = cfg.vocos_dim, F = cfg.vocos_intermediate_dim, cd = cfg.codebook_dim, cs = cfg.codebook_size;
  IDX_CHECK(Hs > 0 && (Hs & 31) == 0 && D > 0 && (D & 3) == 0 && D <= 1024 && F > 0 && (F & 3) == 0, "RepCodec widths");
  IDX_CHECK(cfg.vocos_num_layers > 0 && cd > 0 && (cd & 3) == 0 && cd <= 64 && cs > 0, "RepCodec codebook");
  {   // embed: torch [D][Hs][7] -> rows [D][tap * Hs + ci]
    HostTensor *w = nullptr, *b = nullptr;
    if (need(t, "encoder.0.embed.weight", {D, Hs, 7}, &w) || need(t, "encoder.0.embed.bias", {D}, &b)) return 1;
    std::vector<float> r((size_t)D * 7 * Hs);
    for (int n = 0; n < D; ++n)
      for (int ci = 0; ci < Hs; ++ci)
        for (int k = 0; k < 7; ++k) r[((size_t)n * 7 + k) * Hs + ci] = w->data[((size_t)n * Hs + ci) * 7 + k];
    if (make_linear(arena, r.data(), b->data.data(), D, 7 * Hs, 7 * Hs, &embed)) return 1;
  }
  if (ln_from(t, arena, "encoder.0.norm", D, &norm_g, &norm_b) || ln_from(t, arena, "encoder.0.final_layer_norm", D, &fin_g, &fin_b)) return 1;
  layers.resize(cfg.vocos_num_layers);
  for (int i = 0; i < cfg.vocos_num_layers; ++i) {
    ConvNeXtLayer& L = layers[i];
    const std::string e = "encoder.0.convnext." + std::to_string(i);
    HostTensor *dw = nullptr, *g = nullptr, *w2 = nullptr, *b2 = nullptr;
    if (need(t, e + ".dwconv.weight", {D, 1, 7}, &dw) || up(arena, dw->data, &L.dw_w) || vec_from(t, arena, e + ".dwconv.bias", D, &L.dw_b)) return 1;
    if (ln_from(t, arena, e + ".norm", D, &L.ln_g, &L.ln_b)) return 1;
    if (linear_from(t, arena, e + ".pwconv1", F, D, true, &L.pw1)) return 1;
    if (need(t, e + ".gamma", {D}, &g) || need(t, e + ".pwconv2.weight", {D, F}, &w2) || need(t, e + ".pwconv2.bias", {D}, &b2)) return 1;
    std::vector<float> ws((size_t)D * F), bs(D);      // x = residual + gamma * pwconv2(.)  ->  layer scale folded into pwconv2
    for (int n = 0; n < D; ++n) {
      for (int k = 0; k < F; ++k) ws[(size_t)n * F + k] = g->data[n] * w2->data[(size_t)n * F + k];
      bs[n] = g->data[n] * b2->data[n];
    }
    if (make_linear(arena, ws.data(), bs.data(), D, F, F, &L.pw2)) return 1;
  }
  if (linear_from(t, arena, "encoder.1", Hs, D, true, &enc_out)) return 1;
  const std::string q = "quantizer.quantizers.0";
  if (linear_from(t, arena, q + ".in_project", cd, Hs, true, &in_proj, {cd, Hs, 1})) return 1;
  if (linear_from(t, arena, q + ".out_project", Hs, cd, true, &out_proj, {Hs, cd, 1})) return 1;
  HostTensor* cb = nullptr;
  if (need(t, q + ".codebook.weight", {cs, cd}, &cb) || up(arena, cb->data, &codebook)) return 1;
  std::vector<float> cn(cb->data.size());
  for (int v = 0; v < cs; ++v) {      // F.normalize(codebook): x / max(||x||_2, 1e-12)
    float ss = 0.0f;
    for (int c = 0; c < cd; ++c) ss += cb->data[(size_t)v * cd + c] * cb->data[(size_t)v * cd + c];
    const float nrm = std::max(std::sqrt(ss), 1e-12f);
    for (int c = 0; c < cd; ++c) cn[(size_t)v * cd + c] = cb->data[(size_t)v * cd + c] / nrm;
  }
  return up(arena, cn, &codebook_n);
}

namespace {

struct CodecBuf {
  float *xa, *xb, *h, *ff, *enc, *ze, *zq;
  size_t bytes;
};

CodecBuf carve_codec(const RepCodecModel& m, void* ws, int B, int T) {
  const auto& c = m.cfg;
  const size_t M = (size_t)B * T;
  CodecBuf b;
  Carver k(ws);
  b.xa = k.take<float>(M * c.vocos_dim);
  b.xb = k.take<float>(M * c.vocos_dim);
  b.h = k.take<float>(M * c.vocos_dim);
  b.ff = k.take<float>(M * c.vocos_intermediate_dim);
  b.enc = k.take<float>(M * c.hidden_size);
  b.ze = k.take<float>(M * c.codebook_dim);
  b.zq = k.take<float>(M * c.codebook_dim);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

// decode_latents (factorized_vector_quantize.py:97-119) for one frame per workgroup:
//   e = z / max(||z||, 1e-12);  dist_v = sum(e^2) - 2 e . c_v + sum(c_v^2)  (c_v: L2-normalised code);  index = argmax(-dist), first
//   maximum on ties;  zq = z + (codebook[index] - z)   (the straight-through expression, evaluated as written)
constexpr int CODE_DIM_MAX = 64;
__global__ __launch_bounds__(256) void nearest_code_kernel(const float* ze, const float* codebook, const float* codebook_n, int cs, int cd,
                                                           long long* indices, float* zq) {
  __shared__ float e[CODE_DIM_MAX];
  __shared__ float rv[4];
  __shared__ int ri[4];
  __shared__ float s_e2;
  const int m = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    float ss = 0.0f;
    for (int c = 0; c < cd; ++c) ss += ze[(size_t)m * cd + c] * ze[(size_t)m * cd + c];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float e2 = 0.0f;
    for (int c = 0; c < cd; ++c) { const float v = ze[(size_t)m * cd + c] / nrm; e[c] = v; e2 += v * v; }
    s_e2 = e2;
  }
  __syncthreads();
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int v = tid; v < cs; v += 256) {
    const float* cv = codebook_n + (size_t)v * cd;
    float dot = 0.0f, c2 = 0.0f;
    for (int c = 0; c < cd; ++c) { dot = fmaf(e[c], cv[c], dot); c2 += cv[c] * cv[c]; }
    const float nd = -((s_e2 - 2.0f * dot) + c2);
    if (nd > best) { best = nd; bidx = v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bidx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (rv[w] > best || (rv[w] == best && ri[w] < bidx)) { best = rv[w]; bidx = ri[w]; }
    indices[m] = bidx;
    ri[0] = bidx;
  }
  __syncthreads();
  const int idx = ri[0];
  for (int c = tid; c < cd; c += 256) {
    const float z = ze[(size_t)m * cd + c];
    zq[(size_t)m * cd + c] = z + (codebook[(size_t)idx * cd + c] - z);
  }
}

}  // namespace

size_t RepCodecModel::workspace_bytes(int B, int T) const { return carve_codec(*this, nullptr, B, T).bytes; }

int RepCodecModel::quantize(const float* x, int B, int T, long long* indices, float* s_out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(x && indices && s_out, "null pointer");
  IDX_CHECK(B > 0 && T > 0, "shape");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, T), "workspace too small");
  const int Hs = cfg.hidden_size, D = cfg.vocos_dim, F = cfg.vocos_intermediate_dim, cd = cfg.codebook_dim, M = B * T;
  CodecBuf w = carve_codec(*this, ws, B, T);
  auto ln = [&](const float* in, float* o, const float* g, const float* b) {
    RowsNormArgs n;
    n.x_in = in; n.ld_in = D; n.y = o; n.ld_y = D; n.M = M; n.d = D; n.mode = NORM_LN; n.eps = 1e-6f; n.g1 = g; n.b1 = b;
    return rows_norm_forward(n, st);
  };
  {   // embed: Conv1d(k7, padding 3) over each sequence of T rows
    GemmArgs g;
    g.x = x; g.ldx = Hs; g.y = w.h; g.ldy = D; g.M = M; g.taps = 7; g.seq_len = T; g.dil = 1; g.pad_left = 3; g.pad_mode = 0;
    if (gemm_tn_forward(embed, g, st)) return 1;
  }
  if (ln(w.h, w.xa, norm_g, norm_b)) return 1;
  float* cur = w.xa;
  float* nxt = w.xb;
  for (const ConvNeXtLayer& L : layers) {
    if (dwconv_ln(w.h, cur, L.dw_w, L.dw_b, L.ln_g, L.ln_b, B, T, D, 7, 1e-6f, st)) return 1;
    if (lin_exact(L.pw1, w.h, D, w.ff, F, M, st, ACT_GELU_ERF)) return 1;
    if (lin_exact(L.pw2, w.ff, F, nxt, D, M, st, ACT_NONE, cur, D)) return 1;
    std::swap(cur, nxt);
  }
  if (ln(cur, w.h, fin_g, fin_b)) return 1;
  if (lin_exact(enc_out, w.h, D, w.enc, Hs, M, st)) return 1;
  if (lin_exact(in_proj, w.enc, Hs, w.ze, cd, M, st)) return 1;
  hipLaunchKernelGGL(nearest_code_kernel, dim3(M), dim3(256), 0, st, w.ze, codebook, codebook_n, cfg.codebook_size, cd, indices, w.zq);
  IDX_LAUNCH_CHECK();
  return lin_exact(out_proj, w.zq, cd, s_out, Hs, M, st);
}

}  // namespace idxtts
