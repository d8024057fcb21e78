// w2v-bert-2.0 semantic features of a prompt on MI355X (the `get_emb` call of the reference's prompt block).
//
// Reference: IndexTTS2.get_emb                     indextts/infer_v2.py:381-408  (hidden_states[17], (x - mean) / std)
//            build_semantic_model                  indextts/utils/maskgct_utils.py:87-93  (facebook/w2v-bert-2.0 + wav2vec2bert_stats.pt)
//            the model itself is a third-party dependency (transformers, pinned 4.52.1 by the reference): Wav2Vec2BertModel =
//            feature_projection (LayerNorm 160 + Linear 160 -> 1024) + conformer layers
//              x = x + 0.5 ffn1(LN x);  x = x + attn(LN x) [relative_key distance embedding, 64 left / 8 right];
//              x = x + conv(LN x) [pointwise 2D GLU, CAUSAL depthwise k31, LayerNorm, swish, pointwise];  x = LN(x + 0.5 ffn2(LN x))
//            (modeling_wav2vec2_bert.py: Wav2Vec2BertEncoderLayer / SelfAttention / ConvolutionModule).
//
// Runs once per prompt (the reference caches its result per prompt file, infer_v2.py:618-660).  Token-major rows
// [B*T][channels], every linear on the exact-fp32 MFMA GEMM, attention one wave per query (cond_ops.hip) -- 750 frames for a
// 15 s prompt.  Padded frames (t >= len[b]) are masked as attention keys and as depthwise-conv inputs, exactly where the
// reference masks them; the values the reference computes AT padded frames are not reproduced (they never reach a valid
// frame: the depthwise conv is causal and the padding sits on the right).
#include <cmath>

#include "model_util.h"
#include "semantic.h"
#include "attention.h"

namespace idxtts {

bool W2VBertModel::accepts(const std::string& name) const {
  return name.rfind("feature_projection.", 0) == 0 || name.rfind("encoder.layers.", 0) == 0 || name == "semantic_mean" || name == "semantic_std";
}

int W2VBertModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int D = cfg.hidden_size, H = cfg.num_heads, F = cfg.intermediate_size, In = cfg.input_dim, kc = cfg.conv_kernel;
  IDX_CHECK(D > 0 && H > 0 && D % H == 0 && (D & 3) == 0 && D <= 1024, "hidden_size: a multiple of 4 and of num_heads, at most 1024");
  IDX_CHECK(In > 0 && F > 0 && (F & 3) == 0 && cfg.num_layers > 0 && (kc & 1) == 1 && cfg.left_max >= 0 && cfg.right_max >= 0, "w2v-bert shape");
  dk = D / H;
  IDX_CHECK((dk & 3) == 0 && dk <= 128, "head_dim must be a multiple of 4, at most 128");
  if (ln_from(t, arena, "feature_projection.layer_norm", In, &fp_g, &fp_b)) return 1;
  if (linear_from(t, arena, "feature_projection.projection", D, In, true, &proj)) return 1;
  layers.resize(cfg.num_layers);
  const int nd = cfg.left_max + cfg.right_max + 1;
  for (int i = 0; i < cfg.num_layers; ++i) {
    W2VLayer& L = layers[i];
    const std::string e = "encoder.layers." + std::to_string(i);
    if (ln_from(t, arena, e + ".ffn1_layer_norm", D, &L.ffn1_g, &L.ffn1_b) || ln_from(t, arena, e + ".self_attn_layer_norm", D, &L.att_g, &L.att_b) ||
        ln_from(t, arena, e + ".conv_module.layer_norm", D, &L.conv_g, &L.conv_b) ||
        ln_from(t, arena, e + ".conv_module.depthwise_layer_norm", D, &L.dwn_g, &L.dwn_b) ||
        ln_from(t, arena, e + ".ffn2_layer_norm", D, &L.ffn2_g, &L.ffn2_b) || ln_from(t, arena, e + ".final_layer_norm", D, &L.fin_g, &L.fin_b)) return 1;
    if (linear_from(t, arena, e + ".ffn1.intermediate_dense", F, D, true, &L.ffn1_in) || linear_from(t, arena, e + ".ffn1.output_dense", D, F, true, &L.ffn1_out) ||
        linear_from(t, arena, e + ".ffn2.intermediate_dense", F, D, true, &L.ffn2_in) || linear_from(t, arena, e + ".ffn2.output_dense", D, F, true, &L.ffn2_out)) return 1;
    {   // q, k, v projections stacked into one [3D][D] GEMM
      std::vector<float> w((size_t)3 * D * D), b((size_t)3 * D);
      const char* names[3] = {"linear_q", "linear_k", "linear_v"};
      for (int s = 0; s < 3; ++s) {
        HostTensor *lw = nullptr, *lb = nullptr;
        if (need(t, e + ".self_attn." + names[s] + ".weight", {D, D}, &lw) || need(t, e + ".self_attn." + names[s] + ".bias", {D}, &lb)) return 1;
        std::memcpy(&w[(size_t)s * D * D], lw->data.data(), (size_t)D * D * sizeof(float));
        std::memcpy(&b[(size_t)s * D], lb->data.data(), D * sizeof(float));
      }
      if (make_linear(arena, w.data(), b.data(), 3 * D, D, D, &L.qkv)) return 1;
    }
    if (linear_from(t, arena, e + ".self_attn.linear_out", D, D, true, &L.out)) return 1;
    HostTensor *de = nullptr, *dw = nullptr;
    if (need(t, e + ".self_attn.distance_embedding.weight", {nd, dk}, &de) || up(arena, de->data, &L.dist)) return 1;
    if (linear_from(t, arena, e + ".conv_module.pointwise_conv1", 2 * D, D, false, &L.pw1, {2 * D, D, 1})) return 1;
    if (linear_from(t, arena, e + ".conv_module.pointwise_conv2", D, D, false, &L.pw2, {D, D, 1})) return 1;
    if (need(t, e + ".conv_module.depthwise_conv.weight", {D, 1, kc}, &dw) || up(arena, dw->data, &L.dw_w)) return 1;
  }
  if (t.count("semantic_mean") || t.count("semantic_std")) {
    HostTensor *m = nullptr, *sd = nullptr;
    if (need(t, "semantic_mean", {D}, &m) || need(t, "semantic_std", {D}, &sd)) return 1;
    std::vector<float> inv(D);
    for (int i = 0; i < D; ++i) inv[i] = 1.0f / sd->data[i];
    if (up(arena, m->data, &mean) || up(arena, inv, &inv_std)) return 1;
  }
  return 0;
}

namespace {

struct W2VBuf {
  float *fn, *xa, *xb, *h, *h2, *qkv, *att, *pw, *ff;
  int* len;
  size_t bytes;
};

W2VBuf carve_w2v(const W2VBertModel& m, void* ws, int B, int T) {
  const auto& c = m.cfg;
  const size_t M = (size_t)B * T;
  const int D = c.hidden_size, In4 = (c.input_dim + 3) & ~3;
  W2VBuf b;
  Carver k(ws);
  b.fn = k.take<float>(M * In4);
  b.xa = k.take<float>(M * D);
  b.xb = k.take<float>(M * D);
  b.h = k.take<float>(M * D);
  b.h2 = k.take<float>(M * D);
  b.qkv = k.take<float>(M * 3 * D);
  b.att = k.take<float>(M * D);
  b.pw = k.take<float>(M * 2 * D);
  b.ff = k.take<float>(M * c.intermediate_size);
  b.len = k.take<int>(B);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

// out[m][c] = (x[m][c] - mean[c]) * inv_std[c]
__global__ __launch_bounds__(256) void standardize_rows_kernel(float* out, const float* x, const float* mean, const float* inv_std, int d) {
  const int m = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float v = x[(size_t)m * d + c];
    out[(size_t)m * d + c] = mean ? (v - mean[c]) * inv_std[c] : v;
  }
}

}  // namespace

size_t W2VBertModel::workspace_bytes(int B, int T) const { return carve_w2v(*this, nullptr, B, T).bytes; }

int W2VBertModel::forward(const float* feats, const int* lens_host, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(feats && out, "null pointer");
  IDX_CHECK(B > 0 && T > 0, "shape");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, T), "workspace too small");
  const int D = cfg.hidden_size, H = cfg.num_heads, F = cfg.intermediate_size, In = cfg.input_dim, In4 = (In + 3) & ~3, M = B * T;
  IDX_CHECK(In4 == In, "input_dim must be a multiple of 4");
  W2VBuf w = carve_w2v(*this, ws, B, T);
  std::vector<int> len(B, T);
  bool ragged = false;
  for (int b = 0; b < B; ++b) {
    if (lens_host) {
      IDX_CHECK(lens_host[b] > 0, "a prompt needs at least one valid frame");
      len[b] = std::min(lens_host[b], T);
    }
    ragged = ragged || len[b] != T;
  }
  IDX_HIP(hipMemcpyAsync(w.len, len.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));      // the host vector goes out of scope
  // feature_projection; padded frames start at zero (encoder: hidden_states.masked_fill(~attention_mask, 0))
  RowsNormArgs n0;
  n0.x_in = feats; n0.ld_in = In; n0.y = w.fn; n0.ld_y = In; n0.M = M; n0.d = In; n0.mode = NORM_LN; n0.eps = cfg.layer_norm_eps; n0.g1 = fp_g; n0.b1 = fp_b;
  if (rows_norm_forward(n0, st)) return 1;
  if (lin(proj, w.fn, In, w.xa, D, M, st)) return 1;
  if (ragged && mask_rows(w.xa, M, D, T, w.len, st)) return 1;
  float* x = w.xa;
  float* y = w.xb;
  const float att_scale = 1.0f / std::sqrt((float)dk);
  auto ln = [&](const float* in, float* o, const float* g, const float* b) {
    RowsNormArgs n;
    n.x_in = in; n.ld_in = D; n.y = o; n.ld_y = D; n.M = M; n.d = D; n.mode = NORM_LN; n.eps = cfg.layer_norm_eps; n.g1 = g; n.b1 = b;
    return rows_norm_forward(n, st);
  };
  auto half_ffn = [&](const LinearWeights& w1, const LinearWeights& w2, const float* g, const float* b, const float* xin, float* xout) {
    if (ln(xin, w.h, g, b)) return 1;
    if (lin(w1, w.h, D, w.ff, F, M, st, ACT_SILU)) return 1;
    GemmArgs a;
    a.x = w.ff; a.ldx = F; a.y = xout; a.ldy = D; a.M = M; a.res = xin; a.ldr = D; a.out_scale = 0.5f;      // hidden * 0.5 + residual
    return gemm_forward(w2, a, st);
  };
  for (const W2VLayer& L : layers) {
    if (half_ffn(L.ffn1_in, L.ffn1_out, L.ffn1_g, L.ffn1_b, x, y)) return 1;
    std::swap(x, y);
    // self-attention with the relative_key distance embedding
    if (ln(x, w.h, L.att_g, L.att_b)) return 1;
    if (lin(L.qkv, w.h, D, w.qkv, 3 * D, M, st)) return 1;
    if (dk == 64 && cfg.left_max + cfg.right_max + 1 <= 96) {
      // the MFMA flash kernel with the distance term from an LDS table (attention.h): a 15 s prompt is 750 frames x 16 heads -- the
      // one-wave-per-query kernel below took 0.7 ms per layer for it
      AttnArgs fa;
      fa.q = w.qkv; fa.k = w.qkv + D; fa.v = w.qkv + 2 * D; fa.o = w.att;
      fa.q_bs = fa.k_bs = fa.v_bs = (long)T * 3 * D; fa.o_bs = (long)T * D;
      fa.q_ts = fa.k_ts = fa.v_ts = 3 * D; fa.o_ts = D;
      fa.B = B; fa.H = H; fa.Sq = T; fa.Sk = T; fa.causal = 0; fa.kend = w.len; fa.scale = att_scale;
      fa.rel_key = L.dist; fa.rel_left = cfg.left_max; fa.rel_right = cfg.right_max;
      fa.split_bf16 = get_gemm_mode() == GEMM_BF16X3;
      if (flash_attn_forward(fa, st)) return 1;
    } else {
      SeqAttnArgs a;
      a.q = w.qkv; a.k = w.qkv + D; a.v = w.qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; a.q_bs = a.k_bs = a.v_bs = (long)T * 3 * D;
      a.o = w.att; a.ldo = D; a.o_bs = (long)T * D; a.kend = w.len; a.B = B; a.H = H; a.Sq = T; a.Sk = T; a.dk = dk; a.scale = att_scale;
      a.rel_key = L.dist; a.rel_left = cfg.left_max; a.rel_right = cfg.right_max;
      if (seq_attn_forward(a, st)) return 1;
    }
    if (lin(L.out, w.att, D, y, D, M, st, ACT_NONE, x, D)) return 1;
    std::swap(x, y);
    // convolution module
    if (ln(x, w.h, L.conv_g, L.conv_b)) return 1;
    if (ragged && mask_rows(w.h, M, D, T, w.len, st)) return 1;
    if (lin(L.pw1, w.h, D, w.pw, 2 * D, M, st)) return 1;
    if (glu_dwconv_ln_silu(w.h2, w.pw, L.dw_w, nullptr, L.dwn_g, L.dwn_b, B, T, D, cfg.conv_kernel, st, cfg.conv_kernel - 1)) return 1;
    if (lin(L.pw2, w.h2, D, y, D, M, st, ACT_NONE, x, D)) return 1;
    std::swap(x, y);
    if (half_ffn(L.ffn2_in, L.ffn2_out, L.ffn2_g, L.ffn2_b, x, y)) return 1;
    if (ln(y, x, L.fin_g, L.fin_b)) return 1;
  }
  hipLaunchKernelGGL(standardize_rows_kernel, dim3(M), dim3(256), 0, st, out, x, mean, inv_std, D);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
