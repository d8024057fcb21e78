// Prompt-conditioning encoders of IndexTTS-2's GPT stage on MI355X: ConformerEncoder + PerceiverResampler, and the
// emotion-vector head.
//
// Reference: UnifiedVoice.get_conditioning / get_emo_conditioning / get_emovec  indextts/gpt/model_v2.py:627-671, 897-902
//            ConformerEncoder            indextts/gpt/conformer_encoder.py:284-520 (layer 219-281, conv module 57-164)
//            Conv2dSubsampling2          indextts/gpt/conformer/subsampling.py:131-181
//            RelPositionMultiHeadedAttention   indextts/gpt/conformer/attention.py:164-312 (no rel_shift in this fork)
//            PerceiverResampler          indextts/gpt/perceiver.py:193-317
//
// Runs once per prompt (hoisted out of the segment loop, where the reference recomputes it for every segment:
// infer_v2.py:748-765).  Token-major rows [B*T'][channels] throughout, every linear on the exact-fp32 MFMA GEMM
// (gemm.hip); the one large contraction -- Conv2dSubsampling2's Linear(D*511 -> D), K = 261 632, 134 M weights, a few
// hundred rows -- runs split-K over ~512 workgroups into partial slabs that the next kernel sums in a fixed order.
// Rows beyond a prompt's own length (ragged batches) are masked where the reference masks them: as attention keys, as
// depthwise-conv inputs and as perceiver context; their own values are never read by a valid row.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "cond.h"
#include "model_util.h"

namespace idxtts {

CondModel::CondModel(const idxtts_cond_config& c) : cfg(c) {
  cprefix = c.emotion ? "emo_conditioning_encoder." : "conditioning_encoder.";
  pprefix = c.emotion ? "emo_perceiver_encoder." : "perceiver_encoder.";
}

bool CondModel::accepts(const std::string& name) const {
  if (name.rfind(cprefix, 0) == 0 || name.rfind(pprefix, 0) == 0) return true;
  return cfg.emotion && (name.rfind("emovec_layer.", 0) == 0 || name.rfind("emo_layer.", 0) == 0);
}

int CondModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int D = cfg.output_size, H = cfg.attention_heads, LU = cfg.linear_units, kc = cfg.cnn_kernel;
  IDX_CHECK(D > 0 && H > 0 && D % H == 0 && (D & 3) == 0, "output_size must be a multiple of 4 and of attention_heads");
  IDX_CHECK(cfg.input_size >= 3 && cfg.num_blocks > 0 && (kc & 1) == 1, "conformer shape");
  IDX_CHECK(cfg.perceiver_dim > 0 && (cfg.perceiver_dim & 3) == 0 && cfg.num_latents > 0 && cfg.perceiver_depth > 0, "perceiver shape");
  IDX_CHECK(cfg.perceiver_dim_head > 0 && (cfg.perceiver_dim_head & 3) == 0 && cfg.perceiver_dim_head <= 128, "perceiver head_dim");
  dk = D / H;
  IDX_CHECK((dk & 3) == 0 && dk <= 128, "conformer head_dim must be a multiple of 4, at most 128");
  F2 = (cfg.input_size - 1) / 2;
  const int dim = cfg.perceiver_dim;
  inner = cfg.perceiver_dim_head * H;
  ffi = (int)(dim * cfg.perceiver_mult * 2 / 3);      // FeedForward: int(dim * mult * 2 / 3)  (perceiver.py:181)
  ffi_pad = (ffi + 3) & ~3;
  const std::string c = cprefix.substr(0, cprefix.size() - 1), p = pprefix.substr(0, pprefix.size() - 1);
  // ---- Conv2dSubsampling2 + RelPositionalEncoding ----
  HostTensor *cw = nullptr, *cb = nullptr, *ow = nullptr, *ob = nullptr;
  if (need(t, c + ".embed.conv.0.weight", {D, 1, 3, 3}, &cw) || need(t, c + ".embed.conv.0.bias", {D}, &cb)) return 1;
  if (up(arena, cw->data, &sub_w) || up(arena, cb->data, &sub_b)) return 1;
  if (need(t, c + ".embed.out.0.weight", {D, (int64_t)D * F2}, &ow) || need(t, c + ".embed.out.0.bias", {D}, &ob)) return 1;
  {
    // x * xscale (embedding.py:137) folded into the projection: W' = sqrt(D) W, b' = sqrt(D) b
    const float s = std::sqrt((float)D);
    std::vector<float> ws(ow->data.size()), bs(D);
    for (size_t i = 0; i < ws.size(); ++i) ws[i] = ow->data[i] * s;
    for (int i = 0; i < D; ++i) bs[i] = ob->data[i] * s;
    if (make_linear(arena, ws.data(), nullptr, D, D * F2, D * F2, &embed)) return 1;
    if (up(arena, bs, &embed_bias)) return 1;
  }
  {
    auto it = t.find(c + ".embed.pos_enc.pe");       // the reference's registered buffer [1][max_len][D] (embedding.py:45-53)
    if (it == t.end()) IDX_FAIL("missing tensor '" + c + ".embed.pos_enc.pe' (the sinusoid buffer of the reference's state_dict)");
    const auto& sh = it->second.shape;
    IDX_CHECK(sh.size() == 3 && sh[0] == 1 && sh[2] == D && sh[1] > 0, "pos_enc.pe must be [1][max_len][output_size]");
    pe_len = (int)sh[1];
    if (up(arena, it->second.data, &pe)) return 1;
  }
  if (ln_from(t, arena, c + ".after_norm", D, &after_g, &after_b)) return 1;
  layers.resize(cfg.num_blocks);
  for (int i = 0; i < cfg.num_blocks; ++i) {
    ConformerLayer& L = layers[i];
    const std::string e = c + ".encoders." + std::to_string(i);
    if (ln_from(t, arena, e + ".norm_mha", D, &L.mha_g, &L.mha_b) || ln_from(t, arena, e + ".norm_conv", D, &L.conv_g, &L.conv_b) ||
        ln_from(t, arena, e + ".norm_ff", D, &L.ff_g, &L.ff_b) || ln_from(t, arena, e + ".norm_final", D, &L.fin_g, &L.fin_b)) return 1;
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
    if (linear_from(t, arena, e + ".self_attn.linear_pos", D, D, false, &L.pos)) return 1;
    HostTensor *bu = nullptr, *bv = nullptr;
    if (need(t, e + ".self_attn.pos_bias_u", {H, dk}, &bu) || need(t, e + ".self_attn.pos_bias_v", {H, dk}, &bv)) return 1;
    if (up(arena, bu->data, &L.bias_u) || up(arena, bv->data, &L.bias_v)) return 1;
    if (linear_from(t, arena, e + ".conv_module.pointwise_conv1", 2 * D, D, true, &L.pw1, {2 * D, D, 1})) return 1;
    if (linear_from(t, arena, e + ".conv_module.pointwise_conv2", D, D, true, &L.pw2, {D, D, 1})) return 1;
    HostTensor* dw = nullptr;
    if (need(t, e + ".conv_module.depthwise_conv.weight", {D, 1, kc}, &dw) || up(arena, dw->data, &L.dw_w)) return 1;
    if (vec_from(t, arena, e + ".conv_module.depthwise_conv.bias", D, &L.dw_b)) return 1;
    if (ln_from(t, arena, e + ".conv_module.norm", D, &L.dwn_g, &L.dwn_b)) return 1;
    if (linear_from(t, arena, e + ".feed_forward.w_1", LU, D, true, &L.ff1)) return 1;
    if (linear_from(t, arena, e + ".feed_forward.w_2", D, LU, true, &L.ff2)) return 1;
  }
  // ---- PerceiverResampler ----
  if (dim != D) {
    if (linear_from(t, arena, p + ".proj_context", dim, D, true, &proj_ctx)) return 1;
  } else {
    IDX_CHECK(t.find(p + ".proj_context.weight") == t.end(), "proj_context is nn.Identity when dim_context == dim");
  }
  HostTensor* lat = nullptr;
  if (need(t, p + ".latents", {cfg.num_latents, dim}, &lat) || up(arena, lat->data, &latents)) return 1;
  player.resize(cfg.perceiver_depth);
  for (int l = 0; l < cfg.perceiver_depth; ++l) {
    PerceiverLayer& P = player[l];
    const std::string a = p + ".layers." + std::to_string(l);
    if (linear_from(t, arena, a + ".0.to_q", inner, dim, false, &P.to_q) || linear_from(t, arena, a + ".0.to_kv", 2 * inner, dim, false, &P.to_kv) ||
        linear_from(t, arena, a + ".0.to_out", dim, inner, false, &P.to_out)) return 1;
    if (linear_from(t, arena, a + ".1.0", 2 * ffi, dim, true, &P.ff1) || linear_from(t, arena, a + ".1.2", dim, ffi, true, &P.ff2)) return 1;
  }
  if (vec_from(t, arena, p + ".norm.gamma", dim, &pnorm_g)) return 1;
  if (cfg.emotion) {
    IDX_CHECK(cfg.model_dim > 0 && (cfg.model_dim & 3) == 0, "model_dim");
    if (linear_from(t, arena, "emovec_layer", cfg.model_dim, dim, true, &emovec)) return 1;
    if (linear_from(t, arena, "emo_layer", cfg.model_dim, cfg.model_dim, true, &emo)) return 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
namespace {

struct CondBuf {
  float *a, *slab, *xa, *xb, *h, *h2, *qkv, *att, *pw, *ff, *posp;
  float *ctxp, *lat, *lat2, *cat, *kv, *q, *o, *ffh, *ffg, *ev;
  int *len2, *kend;
  int T2, ksplit;
  size_t bytes;
};

int embed_ksplit(int M, int D, int K) {
  const int ksteps = cdiv(cdiv(K, 16), 2);
  const int tiles = cdiv(M, 128) * cdiv(D, 128);
  int ks = std::max(1, std::min(64, 512 / std::max(1, tiles)));
  ks = std::min(ks, ksteps);
  const int per = cdiv(ksteps, ks);
  return cdiv(ksteps, per);                // no empty K range
}

CondBuf carve(const CondModel& m, void* ws, int B, int T) {
  const auto& c = m.cfg;
  const int D = c.output_size, dim = c.perceiver_dim, n = c.num_latents;
  CondBuf b;
  b.T2 = (T - 3) / 2 + 1;
  const size_t M = (size_t)B * b.T2, Mc = (size_t)B * (n + b.T2), Ml = (size_t)B * n;
  const int K = D * m.F2;
  b.ksplit = embed_ksplit((int)M, D, K);
  Carver k(ws);
  b.a = k.take<float>(M * K);
  b.slab = k.take<float>((size_t)b.ksplit * M * D);
  b.xa = k.take<float>(M * D);
  b.xb = k.take<float>(M * D);
  b.h = k.take<float>(M * D);
  b.h2 = k.take<float>(M * D);
  b.qkv = k.take<float>(M * 3 * D);
  b.att = k.take<float>(M * D);
  b.pw = k.take<float>(M * 2 * D);
  b.ff = k.take<float>(M * c.linear_units);
  b.posp = k.take<float>((size_t)b.T2 * D);
  b.ctxp = k.take<float>(M * dim);
  b.lat = k.take<float>(Ml * dim);
  b.lat2 = k.take<float>(Ml * dim);
  b.cat = k.take<float>(Mc * dim);
  b.kv = k.take<float>(Mc * 2 * m.inner);
  b.q = k.take<float>(Ml * m.inner);
  b.o = k.take<float>(Ml * m.inner);
  b.ffh = k.take<float>(Ml * 2 * m.ffi);
  b.ffg = k.take<float>(Ml * m.ffi_pad);
  b.ev = k.take<float>((size_t)B * std::max(c.model_dim, dim));
  b.len2 = k.take<int>(B);
  b.kend = k.take<int>(B);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

}  // namespace

size_t CondModel::workspace_bytes(int B, int T) const { return carve(*this, nullptr, B, T).bytes; }

int CondModel::forward(const float* feats, const int* lens_host, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(feats && out, "null pointer");
  IDX_CHECK(B > 0 && T >= 3, "Conv2dSubsampling2 needs at least 3 frames");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, T), "workspace too small");
  const int D = cfg.output_size, H = cfg.attention_heads, dim = cfg.perceiver_dim, n = cfg.num_latents;
  CondBuf w = carve(*this, ws, B, T);
  const int T2 = w.T2, M = B * T2, Ml = B * n;
  IDX_CHECK(T2 <= pe_len, "prompt longer than the positional-encoding table");
  // subsampled valid lengths: mask[:, :, 2::2] of (t < len)  (subsampling.py:181); len > T means "no padding" (the reference
  // passes shape[-1] = 1024 here, infer_v2.py:751-752)
  std::vector<int> len2(B, T2), kend(B);
  bool ragged = false;
  for (int b = 0; b < B; ++b) {
    if (lens_host) {
      IDX_CHECK(lens_host[b] >= 3, "a prompt needs at least 3 valid frames");
      const int l = std::min(lens_host[b], T);
      len2[b] = std::min(T2, (l - 3) / 2 + 1);
    }
    ragged = ragged || len2[b] != T2;
    kend[b] = n + len2[b];
  }
  IDX_HIP(hipMemcpyAsync(w.len2, len2.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.kend, kend.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));      // the host vectors go out of scope
  // ---- Conv2dSubsampling2: conv + ReLU -> [M][D*F2]; Linear (split-K) ; x * sqrt(D) folded in ----
  if (sub2_conv_relu(w.a, feats, sub_w, sub_b, B, T, cfg.input_size, D, st)) return 1;
  {
    GemmArgs g;
    g.x = w.a; g.ldx = D * F2; g.y = w.slab; g.ldy = D; g.M = M; g.ksplit = w.ksplit;
    if (gemm_tn_forward(embed, g, st)) return 1;
    RowsNormArgs r;
    r.partials = w.slab; r.num_partials = w.ksplit; r.partial_rows = M; r.ld_partial = D; r.add_bias = embed_bias;
    r.y = w.xa; r.ld_y = D; r.M = M; r.d = D; r.mode = NORM_NONE;
    if (rows_norm_forward(r, st)) return 1;
  }
  float* x = w.xa;
  float* y = w.xb;
  const float att_scale = 1.0f / std::sqrt((float)dk);
  for (const ConformerLayer& L : layers) {
    // rel-pos self-attention
    if (layer_norm(x, w.h, L.mha_g, L.mha_b, M, D, st)) return 1;
    if (lin(L.qkv, w.h, D, w.qkv, 3 * D, M, st)) return 1;
    if (lin(L.pos, pe, D, w.posp, D, T2, st)) return 1;
    SeqAttnArgs a;
    a.q = w.qkv; a.k = w.qkv + D; a.v = w.qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; a.q_bs = a.k_bs = a.v_bs = (long)T2 * 3 * D;
    a.pos = w.posp; a.ldp = D; a.bias_u = L.bias_u; a.bias_v = L.bias_v;
    a.o = w.att; a.ldo = D; a.o_bs = (long)T2 * D; a.kend = w.len2; a.B = B; a.H = H; a.Sq = T2; a.Sk = T2; a.dk = dk; a.scale = att_scale;
    if (seq_attn_forward(a, st)) return 1;
    if (lin(L.out, w.att, D, y, D, M, st, ACT_NONE, x, D)) return 1;
    std::swap(x, y);
    // convolution module
    if (layer_norm(x, w.h, L.conv_g, L.conv_b, M, D, st)) return 1;
    if (ragged && mask_rows(w.h, M, D, T2, w.len2, st)) return 1;
    if (lin(L.pw1, w.h, D, w.pw, 2 * D, M, st)) return 1;
    if (glu_dwconv_ln_silu(w.h2, w.pw, L.dw_w, L.dw_b, L.dwn_g, L.dwn_b, B, T2, D, cfg.cnn_kernel, st)) return 1;
    if (lin(L.pw2, w.h2, D, y, D, M, st, ACT_NONE, x, D)) return 1;
    std::swap(x, y);
    // feed forward (ff_scale 1: macaron style is off), norm_final
    if (layer_norm(x, w.h, L.ff_g, L.ff_b, M, D, st)) return 1;
    if (lin(L.ff1, w.h, D, w.ff, cfg.linear_units, M, st, ACT_SILU)) return 1;
    if (lin(L.ff2, w.ff, cfg.linear_units, y, D, M, st, ACT_NONE, x, D)) return 1;
    if (layer_norm(y, x, L.fin_g, L.fin_b, M, D, st)) return 1;
  }
  if (layer_norm(x, w.h, after_g, after_b, M, D, st)) return 1;
  // ---- PerceiverResampler ----
  const float* ctx = w.h;
  if (dim != D) {
    if (lin(proj_ctx, w.h, D, w.ctxp, dim, M, st)) return 1;
    ctx = w.ctxp;
  }
  for (int b = 0; b < B; ++b)
    IDX_HIP(hipMemcpyAsync(w.lat + (size_t)b * n * dim, latents, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToDevice, st));
  float* lat = w.lat;
  float* lat2 = w.lat2;
  const int Sk = n + T2;
  for (const PerceiverLayer& P : player) {
    if (concat_latents_ctx(w.cat, lat, ctx, B, n, T2, dim, st)) return 1;
    if (lin(P.to_kv, w.cat, dim, w.kv, 2 * inner, B * Sk, st) || lin(P.to_q, lat, dim, w.q, inner, Ml, st)) return 1;
    SeqAttnArgs a;
    a.q = w.q; a.ldq = inner; a.q_bs = (long)n * inner;
    a.k = w.kv; a.v = w.kv + inner; a.ldk = a.ldv = 2 * inner; a.k_bs = a.v_bs = (long)Sk * 2 * inner;
    a.o = w.o; a.ldo = inner; a.o_bs = (long)n * inner; a.kend = w.kend;
    a.B = B; a.H = H; a.Sq = n; a.Sk = Sk; a.dk = cfg.perceiver_dim_head; a.scale = 1.0f / std::sqrt((float)cfg.perceiver_dim_head);
    if (seq_attn_forward(a, st)) return 1;
    if (lin(P.to_out, w.o, inner, lat2, dim, Ml, st, ACT_NONE, lat, dim)) return 1;
    if (lin(P.ff1, lat2, dim, w.ffh, 2 * ffi, Ml, st)) return 1;
    if (geglu(w.ffg, ffi_pad, w.ffh, Ml, ffi, st)) return 1;
    if (lin(P.ff2, w.ffg, ffi_pad, lat, dim, Ml, st, ACT_NONE, lat2, dim)) return 1;
  }
  if (!cfg.emotion) return l2norm_scale(out, lat, pnorm_g, Ml, dim, st);
  // get_emovec: emo_layer(emovec_layer(perceiver(...).squeeze(1)))   (model_v2.py:897-902)
  IDX_CHECK(n == 1, "the emotion perceiver has one latent");
  if (l2norm_scale(lat2, lat, pnorm_g, Ml, dim, st)) return 1;
  if (lin(emovec, lat2, dim, w.ev, cfg.model_dim, B, st)) return 1;
  return lin(emo, w.ev, cfg.model_dim, out, cfg.model_dim, B, st);
}

}  // namespace idxtts
