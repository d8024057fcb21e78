// IndexTTS-2 semantic-to-mel stage on MI355X: gpt_layer + vq2emb + length regulator, and the CFM Euler
// solver over the DiT (13 adaLN-RMSNorm/rotary/SwiGLU blocks with U-ViT skips) + WaveNet estimator.
//
// Reference: infer_v2.py:835-856 (caller), commons.py:390-420 (MyModel), length_regulator.py:90-141,
// flow_matching.py:31-115, diffusion_transformer.py:186-257, gpt_fast/model.py:121-360, wavenet.py:138-166.
//
// Layout: everything is token-major [rows = (stacked batch, frame)][channels] so that every linear AND every
// convolution (WaveNet k=5 reflect-pad, length-regulator k=3) is the same fp32-MFMA GEMM (gemm.hip; the conv
// taps are shifted row reads inside the tile loader, no im2col, no transposes between the DiT and the WaveNet).
// Only the 80-channel ODE state x stays channels-first [B][80][T] (the boundary layout, flow_matching.py:52).
//
// What is hoisted out of the Euler loop (all of it depends on t only, not on the data):
//   timestep MLPs, all 27 adaLN modulation vectors, the FinalLayer shift/scale and the WaveNet conditioning
//   bias g_l (folded with the in_layer bias) are computed ONCE per call for all steps by five small GEMMs
//   with M = n_steps; cond_projection(mu) is computed once per call.  The reference recomputes all of them
//   in every step (diffusion_transformer.py:212-213, wavenet.py:142-152).
// CFG: the [cond | null] stacking of flow_matching.py:89-93 is kept (2B rows per step).
#include <cmath>
#include <cstring>

#include <map>
#include <mutex>
#include <utility>

#include "s2mel.h"

namespace idxtts {

S2MelModel::S2MelModel(const idxtts_s2mel_config& c) : cfg(c) {}

bool S2MelModel::accepts(const std::string& name) const {
  static const char* prefixes[] = {"cfm.estimator.", "length_regulator.", "gpt_layer.", "semantic_codec.", "rope_cache"};
  for (const char* p : prefixes)
    if (name.rfind(p, 0) == 0) return true;
  return false;
}

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

// rows [N][K] host matrix (+ optional bias) -> packed LinearWeights
int make_linear(DeviceArena& arena, const std::vector<float>& w, const std::vector<float>* bias, int N, int K, LinearWeights* out) {
  std::vector<float> packed(linear_packed_floats(N, K));
  pack_linear(packed.data(), w.data(), N, K);
  if (up(arena, packed, &out->wp)) return 1;
  out->N = N; out->K = K;
  {   // split-bf16 copy for the M >= 256 launches (gemm_bf16x3.hip)
    std::vector<float> p16((linear_bf16x3_packed_bytes(N, K) + 3) / 4);
    pack_linear_bf16x3(p16.data(), w.data(), N, K);
    const float* d16 = nullptr;
    if (up(arena, p16, &d16)) return 1;
    out->wp16 = d16;
  }
  if (bias && up(arena, *bias, &out->bias)) return 1;
  return 0;
}

int linear_from(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& prefix, int N, int K, bool bias,
                LinearWeights* out) {
  HostTensor *w = nullptr, *b = nullptr;
  if (need(t, prefix + ".weight", {N, K}, &w)) return 1;
  if (bias && need(t, prefix + ".bias", {N}, &b)) return 1;
  return make_linear(arena, w->data, b ? &b->data : nullptr, N, K, out);
}

// columns [k0, k1) of a [N][K] matrix
std::vector<float> col_slice(const std::vector<float>& w, int N, int K, int k0, int k1) {
  std::vector<float> o((size_t)N * (k1 - k0));
  for (int n = 0; n < N; ++n) std::memcpy(&o[(size_t)n * (k1 - k0)], &w[(size_t)n * K + k0], (k1 - k0) * sizeof(float));
  return o;
}

// interleave two [H][K] blocks as [32 of a | 32 of b] groups (SwiGLU / tanh-sigmoid gate packing)
std::vector<float> pair_pack(const float* a, const float* b, int H, int K) {
  std::vector<float> o((size_t)2 * H * K);
  for (int blk = 0; blk < H / 32; ++blk) {
    std::memcpy(&o[(size_t)(blk * 64) * K], a + (size_t)blk * 32 * K, (size_t)32 * K * sizeof(float));
    std::memcpy(&o[(size_t)(blk * 64 + 32) * K], b + (size_t)blk * 32 * K, (size_t)32 * K * sizeof(float));
  }
  return o;
}

// torch Conv1d weight [Cout][Cin][k] -> [Cout][k*Cin] (tap-major K, the order the conv-mode tile loader walks)
std::vector<float> conv_to_rows(const std::vector<float>& w, int Cout, int Cin, int k) {
  std::vector<float> o((size_t)Cout * Cin * k);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int kk = 0; kk < k; ++kk) o[((size_t)co * k + kk) * Cin + ci] = w[((size_t)co * Cin + ci) * k + kk];
  return o;
}

}  // namespace

int S2MelModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int D = cfg.hidden_dim, C = cfg.in_channels, Wh = cfg.wn_hidden, depth = cfg.depth;
  IDX_CHECK(D == cfg.num_heads * 64, "head_dim must be 64");
  IDX_CHECK(Wh == D, "the reference's FinalLayer requires wavenet.hidden_dim == DiT.hidden_dim");
  IDX_CHECK((D & 31) == 0 && (Wh & 31) == 0 && (cfg.lr_channels & 31) == 0, "channel counts must be multiples of 32");
  {
    const int nh = (int)(2 * (4 * D) / 3);
    ffn = (nh % 256 == 0) ? nh : nh + 256 - (nh % 256);
  }
  const std::string e = "cfm.estimator";
  blocks.resize(depth);
  std::vector<float> mod_w, mod_b;
  auto append_mod = [&](const std::string& name) -> int {
    HostTensor *w = nullptr, *b = nullptr;
    if (need(t, name + ".project_layer.weight", {2 * D, D}, &w) || need(t, name + ".project_layer.bias", {2 * D}, &b)) return 1;
    mod_w.insert(mod_w.end(), w->data.begin(), w->data.end());
    mod_b.insert(mod_b.end(), b->data.begin(), b->data.end());
    return 0;
  };
  for (int i = 0; i < depth; ++i) {
    DiTBlock& B = blocks[i];
    const std::string p = e + ".transformer.layers." + std::to_string(i);
    if (linear_from(t, arena, p + ".attention.wqkv", 3 * D, D, false, &B.wqkv)) return 1;
    if (linear_from(t, arena, p + ".attention.wo", D, D, false, &B.wo)) return 1;
    HostTensor *w1 = nullptr, *w3 = nullptr;
    if (need(t, p + ".feed_forward.w1.weight", {ffn, D}, &w1) || need(t, p + ".feed_forward.w3.weight", {ffn, D}, &w3)) return 1;
    if (make_linear(arena, pair_pack(w1->data.data(), w3->data.data(), ffn, D), nullptr, 2 * ffn, D, &B.w13)) return 1;
    if (linear_from(t, arena, p + ".feed_forward.w2", D, ffn, false, &B.w2)) return 1;
    HostTensor *g = nullptr;
    if (need(t, p + ".attention_norm.norm.weight", {D}, &g) || up(arena, g->data, &B.attn_g)) return 1;
    if (need(t, p + ".ffn_norm.norm.weight", {D}, &g) || up(arena, g->data, &B.ffn_g)) return 1;
    if (append_mod(p + ".attention_norm") || append_mod(p + ".ffn_norm")) return 1;
    if (i > depth / 2) {
      HostTensor *sw = nullptr, *sb = nullptr;
      if (need(t, p + ".skip_in_linear.weight", {D, 2 * D}, &sw) || need(t, p + ".skip_in_linear.bias", {D}, &sb)) return 1;
      if (make_linear(arena, col_slice(sw->data, D, 2 * D, 0, D), &sb->data, D, D, &B.skip_a)) return 1;
      if (make_linear(arena, col_slice(sw->data, D, 2 * D, D, 2 * D), nullptr, D, D, &B.skip_b)) return 1;
    }
  }
  {
    HostTensor* g = nullptr;
    if (need(t, e + ".transformer.norm.norm.weight", {D}, &g) || up(arena, g->data, &final_g)) return 1;
    if (append_mod(e + ".transformer.norm")) return 1;
    if (make_linear(arena, mod_w, &mod_b, (2 * depth + 1) * 2 * D, D, &mod_all)) return 1;
  }
  const int Win = 2 * C + D + cfg.style_dim;
  if (linear_from(t, arena, e + ".cond_projection", D, cfg.content_dim, true, &cond_proj)) return 1;
  if (linear_from(t, arena, e + ".cond_x_merge_linear", D, Win, true, &merge)) return 1;
  if (linear_from(t, arena, e + ".t_embedder.mlp.0", D, 256, true, &temb0) || linear_from(t, arena, e + ".t_embedder.mlp.2", D, D, true, &temb2)) return 1;
  if (linear_from(t, arena, e + ".t_embedder2.mlp.0", Wh, 256, true, &t2emb0) || linear_from(t, arena, e + ".t_embedder2.mlp.2", Wh, Wh, true, &t2emb2)) return 1;
  {
    HostTensor *sw = nullptr, *sb = nullptr;
    if (need(t, e + ".skip_linear.weight", {D, D + C}, &sw) || need(t, e + ".skip_linear.bias", {D}, &sb)) return 1;
    if (make_linear(arena, col_slice(sw->data, D, D + C, 0, D), &sb->data, D, D, &skiplin_a)) return 1;
    if (make_linear(arena, col_slice(sw->data, D, D + C, D, D + C), nullptr, D, C, &skiplin_b)) return 1;
  }
  if (linear_from(t, arena, e + ".conv1", Wh, D, true, &conv1)) return 1;
  if (linear_from(t, arena, e + ".res_projection", Wh, D, true, &res_proj)) return 1;
  if (linear_from(t, arena, e + ".final_layer.linear", Wh, Wh, true, &final_lin)) return 1;
  if (linear_from(t, arena, e + ".final_layer.adaLN_modulation.1", 2 * Wh, Wh, true, &final_mod)) return 1;
  {
    HostTensor *w = nullptr, *b = nullptr;
    if (need(t, e + ".conv2.weight", {C, Wh, 1}, &w) || need(t, e + ".conv2.bias", {C}, &b)) return 1;
    if (make_linear(arena, w->data, &b->data, C, Wh, &conv2)) return 1;
  }
  // ---- WaveNet ----
  const int L = cfg.wn_layers, k = cfg.wn_kernel;
  wn.resize(L);
  {
    HostTensor *cw = nullptr, *cb = nullptr;
    if (need(t, e + ".wavenet.cond_layer.conv.conv.weight", {2 * Wh * L, Wh, 1}, &cw) ||
        need(t, e + ".wavenet.cond_layer.conv.conv.bias", {2 * Wh * L}, &cb)) return 1;
    std::vector<float> cw_perm, cb_perm;
    for (int l = 0; l < L; ++l) {
      WNLayer& W = wn[l];
      const std::string p = e + ".wavenet.in_layers." + std::to_string(l) + ".conv.conv";
      HostTensor *iw = nullptr, *ib = nullptr;
      if (need(t, p + ".weight", {2 * Wh, Wh, k}, &iw) || need(t, p + ".bias", {2 * Wh}, &ib)) return 1;
      const std::vector<float> rows = conv_to_rows(iw->data, 2 * Wh, Wh, k);
      if (make_linear(arena, pair_pack(rows.data(), rows.data() + (size_t)Wh * k * Wh, Wh, k * Wh), nullptr, 2 * Wh, k * Wh, &W.in_gate)) return 1;
      // cond_layer slice of this layer, same gate packing; its bias absorbs the in_layer bias
      const float* cwl = cw->data.data() + (size_t)l * 2 * Wh * Wh;
      const std::vector<float> cwp = pair_pack(cwl, cwl + (size_t)Wh * Wh, Wh, Wh);
      cw_perm.insert(cw_perm.end(), cwp.begin(), cwp.end());
      std::vector<float> bsum(2 * Wh);
      for (int c = 0; c < 2 * Wh; ++c) bsum[c] = cb->data[(size_t)l * 2 * Wh + c] + ib->data[c];
      const std::vector<float> bp = pair_pack(bsum.data(), bsum.data() + Wh, Wh, 1);
      cb_perm.insert(cb_perm.end(), bp.begin(), bp.end());
      const std::string r = e + ".wavenet.res_skip_layers." + std::to_string(l) + ".conv.conv";
      const int rc = l < L - 1 ? 2 * Wh : Wh;
      HostTensor *rw = nullptr, *rb = nullptr;
      if (need(t, r + ".weight", {rc, Wh, 1}, &rw) || need(t, r + ".bias", {rc}, &rb)) return 1;
      if (l < L - 1) {
        std::vector<float> w_res(rw->data.begin(), rw->data.begin() + (size_t)Wh * Wh), b_res(rb->data.begin(), rb->data.begin() + Wh);
        std::vector<float> w_skip(rw->data.begin() + (size_t)Wh * Wh, rw->data.end()), b_skip(rb->data.begin() + Wh, rb->data.end());
        if (make_linear(arena, w_res, &b_res, Wh, Wh, &W.res) || make_linear(arena, w_skip, &b_skip, Wh, Wh, &W.skip)) return 1;
      } else {
        W.has_res = false;
        if (make_linear(arena, rw->data, &rb->data, Wh, Wh, &W.skip)) return 1;
      }
    }
    if (make_linear(arena, cw_perm, &cb_perm, 2 * Wh * L, Wh, &wn_cond)) return 1;
  }
  // ---- length regulator, gpt_layer, codec table ----
  const int LC = cfg.lr_channels;
  if (linear_from(t, arena, "length_regulator.content_in_proj", LC, cfg.lr_in_channels, true, &lr_in)) return 1;
  lr_conv.resize(cfg.lr_num_convs);
  lr_gn_g.resize(cfg.lr_num_convs);
  lr_gn_b.resize(cfg.lr_num_convs);
  for (int n = 0; n < cfg.lr_num_convs; ++n) {
    HostTensor *w = nullptr, *b = nullptr, *g = nullptr, *bb = nullptr;
    const std::string p = "length_regulator.model." + std::to_string(3 * n);
    if (need(t, p + ".weight", {LC, LC, 3}, &w) || need(t, p + ".bias", {LC}, &b)) return 1;
    if (make_linear(arena, conv_to_rows(w->data, LC, LC, 3), &b->data, LC, 3 * LC, &lr_conv[n])) return 1;
    const std::string q = "length_regulator.model." + std::to_string(3 * n + 1);
    if (need(t, q + ".weight", {LC}, &g) || need(t, q + ".bias", {LC}, &bb)) return 1;
    if (up(arena, g->data, &lr_gn_g[n]) || up(arena, bb->data, &lr_gn_b[n])) return 1;
  }
  {
    HostTensor *w = nullptr, *b = nullptr;
    const std::string p = "length_regulator.model." + std::to_string(3 * cfg.lr_num_convs);
    if (need(t, p + ".weight", {LC, LC, 1}, &w) || need(t, p + ".bias", {LC}, &b)) return 1;
    if (make_linear(arena, w->data, &b->data, LC, LC, &lr_out)) return 1;
  }
  {
    const int dims[4] = {cfg.gpt_dim, cfg.gpt_layer_dims[0], cfg.gpt_layer_dims[1], cfg.gpt_layer_dims[2]};
    for (int n = 0; n < 3; ++n)
      if (linear_from(t, arena, "gpt_layer." + std::to_string(n), dims[n + 1], dims[n], true, &gl[n])) return 1;
    IDX_CHECK(dims[3] == cfg.codec_hidden && cfg.codec_hidden == cfg.lr_in_channels, "gpt_layer / codec / length-regulator widths");
  }
  {
    // vq2emb table: out_project(codebook[v]) for every v (factorized_vector_quantize.py:123-127)
    HostTensor *cb = nullptr, *ow = nullptr, *ob = nullptr;
    const std::string q = "semantic_codec.quantizer.quantizers.0";
    if (need(t, q + ".codebook.weight", {cfg.codebook_size, cfg.codebook_dim}, &cb) ||
        need(t, q + ".out_project.weight", {cfg.codec_hidden, cfg.codebook_dim, 1}, &ow) ||
        need(t, q + ".out_project.bias", {cfg.codec_hidden}, &ob)) return 1;
    std::vector<float> table((size_t)cfg.codebook_size * cfg.codec_hidden);
    for (int v = 0; v < cfg.codebook_size; ++v)
      for (int h = 0; h < cfg.codec_hidden; ++h) {
        float acc = 0.0f;
        for (int c = 0; c < cfg.codebook_dim; ++c) acc = std::fmaf(cb->data[(size_t)v * cfg.codebook_dim + c], ow->data[(size_t)h * cfg.codebook_dim + c], acc);
        table[(size_t)v * cfg.codec_hidden + h] = acc + ob->data[h];
      }
    if (up(arena, table, &vq_table)) return 1;
  }
  {
    auto it = t.find("rope_cache");
    if (it == t.end()) IDX_FAIL("missing tensor 'rope_cache' ([T][32][2] cos/sin table, precompute_freqs_cis)");
    IDX_CHECK(it->second.shape.size() == 3 && it->second.shape[1] == 32 && it->second.shape[2] == 2, "rope_cache must be [T][32][2]");
    rope_len = (int)it->second.shape[0];
    if (up(arena, it->second.data, &rope)) return 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
struct Carver2 {
  char* base; size_t off = 0;
  explicit Carver2(void* b) : base(static_cast<char*>(b)) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct CfmBuffers {
  float *x_in, *ha, *hb, *hmid, *hn, *qkv, *att, *ff, *xres, *wn_x, *wn_acts, *wn_out, *vout, *condp, *xstate;
  void *hn_p, *att_p, *ff_p, *acts_p;     // split-bf16 planes of hn / att / ff / wn_acts (producers write them for the next GEMM)
  void *h_p, *wnx_p, *xres_p;             // planes of the residual stream entering a skip-receive layer, of wn_x, of xres
  std::vector<void*> skips_p;             // planes of the U-ViT skip tensors
  std::vector<float*> skips;
  float *t1, *t1s, *mods, *fmod, *t2, *wnb, *tmp_steps;
  int *lens2, *plen, *lens2t;      // lens2t: lens2 - tail_t0
  int tail_t0 = 0;                 // first frame the post-transformer part is evaluated on (0 = every frame)
  size_t bytes;
};

static CfmBuffers carve_cfm(const S2MelModel& m, void* ws, int B, int T, int n_steps) {
  const auto& c = m.cfg;
  const int D = c.hidden_dim, C = c.in_channels, Wh = c.wn_hidden;
  const size_t M2 = (size_t)2 * B * T;
  const int Win = 2 * C + D + c.style_dim;
  CfmBuffers b;
  Carver2 k(ws);
  b.x_in = k.take<float>(M2 * Win);
  b.ha = k.take<float>(M2 * D);
  b.hb = k.take<float>(M2 * D);
  b.hmid = k.take<float>(M2 * D);
  b.hn = k.take<float>(M2 * D);
  b.qkv = k.take<float>(M2 * 3 * D);
  b.att = k.take<float>(M2 * D);
  b.ff = k.take<float>(M2 * m.ffn);
  b.xres = k.take<float>(M2 * D);
  b.hn_p = k.take<float>(M2 * std::max(D, Wh));      // 2 bf16 planes = 4 bytes per element
  b.att_p = k.take<float>(M2 * D);
  b.ff_p = k.take<float>(M2 * m.ffn);
  b.acts_p = k.take<float>(M2 * Wh);
  b.h_p = k.take<float>(M2 * D);
  b.wnx_p = k.take<float>(M2 * Wh);
  b.xres_p = k.take<float>(M2 * D);
  b.wn_x = k.take<float>(M2 * Wh);
  b.wn_acts = k.take<float>(M2 * Wh);
  b.wn_out = k.take<float>(M2 * Wh);
  b.vout = k.take<float>(M2 * C);
  b.condp = k.take<float>((size_t)B * T * D);
  b.xstate = k.take<float>((size_t)B * C * T);
  for (int i = 0; i < c.depth / 2; ++i) b.skips.push_back(k.take<float>(M2 * D));
  for (int i = 0; i < c.depth / 2; ++i) b.skips_p.push_back(k.take<float>(M2 * D));
  b.t1 = k.take<float>((size_t)n_steps * D);
  b.t1s = k.take<float>((size_t)n_steps * D);
  b.tmp_steps = k.take<float>((size_t)n_steps * std::max(D, Wh));
  b.mods = k.take<float>((size_t)n_steps * (2 * c.depth + 1) * 2 * D);
  b.fmod = k.take<float>((size_t)n_steps * 2 * Wh);
  b.t2 = k.take<float>((size_t)n_steps * Wh);
  b.wnb = k.take<float>((size_t)n_steps * c.wn_layers * 2 * Wh);
  b.lens2 = k.take<int>(2 * B);
  b.plen = k.take<int>(B);
  b.lens2t = k.take<int>(2 * B);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

size_t S2MelModel::cfm_workspace_bytes(int B, int T, int n_steps) const { return carve_cfm(*this, nullptr, B, T, n_steps).bytes; }

// The buffers of ONE half of the stacked [conditional | null] batch: every activation buffer is [2B*T rows][cols] (or planes
// of that many rows), so half h is the block of B*T rows at h * B*T*cols elements -- for planes too: a half's hi + lo planes
// [cols/16][B*T][16] are B*T*cols*4 bytes, laid out with rows = B*T.  The two halves never read each other's rows between
// cfm_pack and the Euler update, which is what lets them run on two streams (dit_eval_halves).
static CfmBuffers half_view(const S2MelModel& m, const CfmBuffers& w, int h, int B, int T) {
  const auto& c = m.cfg;
  const int D = c.hidden_dim, C = c.in_channels, Wh = c.wn_hidden, Win = 2 * C + D + c.style_dim;
  const size_t R = (size_t)h * B * T;
  CfmBuffers v = w;
  auto sh = [&](float* p, int cols) { return p + R * cols; };
  auto shp = [&](void* p, int cols) -> void* { return static_cast<float*>(p) + R * cols; };
  v.x_in = sh(w.x_in, Win); v.ha = sh(w.ha, D); v.hb = sh(w.hb, D); v.hmid = sh(w.hmid, D); v.hn = sh(w.hn, D); v.qkv = sh(w.qkv, 3 * D);
  v.att = sh(w.att, D); v.ff = sh(w.ff, m.ffn); v.xres = sh(w.xres, D); v.wn_x = sh(w.wn_x, Wh); v.wn_acts = sh(w.wn_acts, Wh);
  v.wn_out = sh(w.wn_out, Wh); v.vout = sh(w.vout, C);
  v.hn_p = shp(w.hn_p, std::max(D, Wh)); v.att_p = shp(w.att_p, D); v.ff_p = shp(w.ff_p, m.ffn); v.acts_p = shp(w.acts_p, Wh);
  v.h_p = shp(w.h_p, D); v.wnx_p = shp(w.wnx_p, Wh); v.xres_p = shp(w.xres_p, D);
  for (size_t i = 0; i < w.skips.size(); ++i) { v.skips[i] = sh(w.skips[i], D); v.skips_p[i] = shp(w.skips_p[i], D); }
  v.lens2 = w.lens2 + h * B; v.lens2t = w.lens2t + h * B;
  return v;
}

// xp / yp: split-bf16 planes of the input / output (GemmArgs::x_planes / y_planes); with xp the fp32 x is not read,
// with yp and y == nullptr no fp32 output is written
static int gemm(const LinearWeights& w, const float* x, int ldx, float* y, int ldy, int M, hipStream_t st, int act = ACT_NONE,
                const float* res = nullptr, int ldr = 0, const void* xp = nullptr, void* yp = nullptr) {
  GemmArgs a;
  a.x = xp ? nullptr : x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.M = M; a.act = act; a.res = res; a.ldr = ldr;
  a.x_planes = xp; a.y_planes = yp;
  return gemm_forward(w, a, st);
}

// N2 sequences of T frames (one CFG half, or the stacked batch); lag_event (optional) is recorded behind the first attention
static int dit_eval(S2MelModel& m, CfmBuffers& w, int N2, int T, int step, hipStream_t st, hipEvent_t lag_event) {
  const auto& c = m.cfg;
  const int D = c.hidden_dim, C = c.in_channels, Wh = c.wn_hidden, depth = c.depth, half = depth / 2;
  const int M = N2 * T, Win = 2 * C + D + c.style_dim;
  const float* mods = w.mods + (size_t)step * (2 * depth + 1) * 2 * D;
  // Producer -> GEMM chaining: when the consumers run on the LDS-DMA split-bf16 kernel, hn / att / ff / wn_acts are
  // written by their producers directly as bf16 hi/lo planes (same bytes as the fp32 row) and never exist in fp32.
  bool chain = true;
  {
    GemmArgs pr;
    pr.M = M;
    const LinearWeights* used[] = {&m.blocks[0].wqkv, &m.blocks[0].wo, &m.blocks[0].w13, &m.blocks[0].w2, &m.skiplin_a, &m.final_lin,
                                   &m.wn[0].skip};
    for (const LinearWeights* lw : used) chain = chain && gemm_uses_planes(*lw, pr);
  }
  bool chain_wn = chain;      // the same for the long skip and the WaveNet operands (xres, wn_x)
  {
    GemmArgs pr;
    pr.M = M;
    const LinearWeights* used[] = {&m.skiplin_b, &m.conv1, &m.res_proj, &m.wn[0].res};
    for (const LinearWeights* lw : used) chain_wn = chain_wn && gemm_uses_planes(*lw, pr);
    pr.taps = c.wn_kernel; pr.seq_len = T;
    chain_wn = chain_wn && gemm_uses_planes(m.wn[0].in_gate, pr);
  }
  void* const hn_p = chain ? w.hn_p : nullptr;
  // The solver needs the estimate only on the frames t >= tail_t0 (see below): the post-transformer part and, in the LAST block,
  // everything after the K/V projection run on those frames alone, compacted to [N2][Tt] rows.
  const int t0 = w.tail_t0, Tt = T - t0, Mt = N2 * Tt;
  bool chain_t = chain, chain_wn_t = chain_wn;
  if (t0 > 0) {
    GemmArgs pr;
    pr.M = Mt;
    const LinearWeights* used[] = {&m.blocks[0].wo, &m.blocks[0].w13, &m.blocks[0].w2, &m.skiplin_a, &m.final_lin, &m.wn[0].skip};
    for (const LinearWeights* lw : used) chain_t = chain_t && gemm_uses_planes(*lw, pr);
    chain_wn_t = chain_t && chain_wn && gemm_uses_planes(m.skiplin_b, pr) && gemm_uses_planes(m.conv1, pr) && gemm_uses_planes(m.res_proj, pr) &&
                 gemm_uses_planes(m.wn[0].res, pr);
    pr.taps = c.wn_kernel; pr.seq_len = Tt;
    chain_wn_t = chain_wn_t && gemm_uses_planes(m.wn[0].in_gate, pr);
  }
  float* const hnt_f = chain_t ? nullptr : w.hn;
  void* const hnt_p = chain_t ? w.hn_p : nullptr;
  auto ada_rows = [&](const float* x, const float* g, int mod_idx, int rows, bool planes) {
    RowsNormArgs n;
    n.x_in = x; n.ld_in = D; n.y = planes ? nullptr : w.hn; n.y_planes = planes ? w.hn_p : nullptr; n.ld_y = D; n.M = rows; n.d = D;
    n.mode = NORM_ADA_RMS; n.eps = c.norm_eps; n.g1 = g;
    n.mod_a = mods + (size_t)mod_idx * 2 * D; n.mod_b = n.mod_a + D; n.ld_mod = 0; n.rows_per_batch = 0;
    return rows_norm_forward(n, st);
  };
  auto ada = [&](const float* x, const float* g, int mod_idx) { return ada_rows(x, g, mod_idx, M, chain); };
  if (lag_event && depth < 2) IDX_HIP(hipEventRecord(lag_event, st));      // (a one-block model has no "behind the first attention")
  bool h_compact = false;      // h already holds only the tail rows (the last block produced it that way)
  if (gemm(m.merge, w.x_in, Win, w.ha, D, M, st)) return 1;
  float* h = w.ha;
  int pushed = 0;
  for (int i = 0; i < depth; ++i) {
    DiTBlock& B = m.blocks[i];
    if (i > half) {     // U-ViT receive: skip_in_linear(cat[x, skip]) as two GEMMs (model.py:233-234)
      float* skip = w.skips[--pushed];
      void* skip_p = chain ? w.skips_p[pushed] : nullptr;
      if (gemm(B.skip_a, h, D, w.hmid, D, M, st, ACT_NONE, nullptr, 0, chain ? w.h_p : nullptr)) return 1;
      float* dst = (h == w.ha) ? w.hb : w.ha;
      if (gemm(B.skip_b, skip, D, dst, D, M, st, ACT_NONE, w.hmid, D, skip_p)) return 1;
      h = dst;
    }
    if (ada(h, B.attn_g, 2 * i)) return 1;
    bool qkv_planes = false;
    {     // qkv projection; the rotary embedding of q and k rides in its epilogue when the LDS-DMA kernel runs it
      GemmArgs g;
      g.x = chain ? nullptr : w.hn; g.x_planes = hn_p; g.ldx = D; g.y = w.qkv; g.ldy = 3 * D; g.M = M;
      const bool fuse_rope = gemm_uses_planes(B.wqkv, g);
      // LDS-DMA GEMM: rotary embedding in its epilogue, and q / k / v leave it as split-bf16 planes (the same bytes as the fp32
      // rows they replace) for the planes attention kernel
      qkv_planes = fuse_rope && D == 64 * c.num_heads;
      if (fuse_rope) { g.rope = m.rope; g.rope_T = T; g.rope_cols = 2 * D; }
      if (qkv_planes) { g.y = nullptr; g.y_planes = w.qkv; }
      if (gemm_forward(B.wqkv, g, st)) return 1;
      if (!fuse_rope && rotary_qk(w.qkv, M, c.num_heads, T, m.rope, st)) return 1;
    }
    AttnPlanesArgs ap;
    ap.planes = w.qkv; ap.Mrows = M; ap.ncols = 3 * D; ap.q_col = 0; ap.k_col = D; ap.v_col = 2 * D; ap.T = T; ap.q_row0 = 0; ap.Sq = T;
    ap.B = N2; ap.H = c.num_heads; ap.kend = w.lens2; ap.scale = 0.125f; ap.o = w.att; ap.o_bs = (long)T * D; ap.o_ts = D;
    AttnArgs a;
    a.q = w.qkv; a.k = w.qkv + D; a.v = w.qkv + 2 * D; a.o = w.att;
    a.q_bs = a.k_bs = a.v_bs = (long)T * 3 * D; a.o_bs = (long)T * D;
    a.q_ts = a.k_ts = a.v_ts = 3 * D; a.o_ts = D;
    a.B = N2; a.H = c.num_heads; a.Sq = T; a.Sk = T; a.causal = 0; a.kend = w.lens2; a.scale = 0.125f;
    a.split_bf16 = get_gemm_mode() == GEMM_BF16X3;
    if (i == depth - 1 && t0 > 0 && i >= half) {
      // last block: its keys / values cover every frame, but only the tail frames' queries, attention output, projection and
      // feed-forward are ever used (nothing attends to this block's output): Mt rows instead of M from here on
      a.q = w.qkv + (size_t)t0 * 3 * D; a.Sq = Tt; a.o_bs = (long)Tt * D;
      ap.q_row0 = t0; ap.Sq = Tt; ap.o_bs = (long)Tt * D;
      if (chain_t) { a.o = nullptr; a.o_planes = w.att_p; ap.o = nullptr; ap.o_planes = w.att_p; }
      if (qkv_planes ? flash_attn_planes_forward(ap, st) : flash_attn_forward(a, st)) return 1;
      if (gather_tail_rows(w.xres, D, h, D, D, N2, T, t0, st)) return 1;                       // residual rows (xres is free here)
      if (gemm(B.wo, w.att, D, w.hmid, D, Mt, st, ACT_NONE, w.xres, D, chain_t ? w.att_p : nullptr)) return 1;
      if (ada_rows(w.hmid, B.ffn_g, 2 * i + 1, Mt, chain_t)) return 1;
      if (gemm(B.w13, w.hn, D, chain_t ? nullptr : w.ff, m.ffn, Mt, st, ACT_SWIGLU, nullptr, 0, hnt_p, chain_t ? w.ff_p : nullptr)) return 1;
      float* dst = (h == w.ha) ? w.hb : w.ha;
      if (gemm(B.w2, w.ff, m.ffn, dst, D, Mt, st, ACT_NONE, w.hmid, D, chain_t ? w.ff_p : nullptr, nullptr)) return 1;
      h = dst;
      h_compact = true;
      break;
    }
    if (chain) { a.o = nullptr; a.o_planes = w.att_p; ap.o = nullptr; ap.o_planes = w.att_p; }
    if (qkv_planes ? flash_attn_planes_forward(ap, st) : flash_attn_forward(a, st)) return 1;
    if (i == 0 && lag_event) IDX_HIP(hipEventRecord(lag_event, st));
    if (gemm(B.wo, w.att, D, w.hmid, D, M, st, ACT_NONE, h, D, chain ? w.att_p : nullptr)) return 1;           // h + attention(...)
    if (ada(w.hmid, B.ffn_g, 2 * i + 1)) return 1;
    if (gemm(B.w13, w.hn, D, chain ? nullptr : w.ff, m.ffn, M, st, ACT_SWIGLU, nullptr, 0, hn_p, chain ? w.ff_p : nullptr)) return 1;
    float* dst = (i < half) ? w.skips[pushed] : ((h == w.ha) ? w.hb : w.ha);
    // its output is read again as a GEMM operand when it is a skip tensor (skip_b later) or enters a skip-receive layer (skip_a)
    void* dst_p = !chain ? nullptr : (i < half ? w.skips_p[pushed] : (i + 1 < depth && i + 1 > half ? w.h_p : nullptr));
    if (gemm(B.w2, w.ff, m.ffn, dst, D, M, st, ACT_NONE, w.hmid, D, chain ? w.ff_p : nullptr, dst_p)) return 1;      // out = h + ffn
    if (i < half) ++pushed;
    h = dst;
  }
  // ---- everything after the transformer is row-local or a short convolution (WaveNet: k = 5, 8 layers -> 16 frames of context),
  // and the Euler step discards the estimate on the prompt frames (x[..., :prompt_len] = 0, flow_matching.py:113): in the solver
  // only the frames t >= tail_t0 = prompt_len - halo are evaluated -- the prompt is 44 % of the frames at configs[2] -- on rows
  // compacted to [N2][T - tail_t0].  The frames t >= prompt_len come out bit-identical: the reflect padding at the cut only reaches
  // the halo.  tail_t0 = 0 (the stand-alone estimator entry point): every frame.
  const float* ht = h;
  const float* xin_t = w.x_in;
  int ld_xin = Win;
  const int* lens_t = w.lens2;
  if (t0 > 0) {
    if (!h_compact) {
      float* hc = (h == w.ha) ? w.hb : w.ha;
      if (gather_tail_rows(hc, D, h, D, D, N2, T, t0, st)) return 1;
      ht = hc;
    }
    if (gather_tail_rows(w.qkv, C, w.x_in, Win, C, N2, T, t0, st)) return 1;      // the x columns of x_in (qkv is free here)
    xin_t = w.qkv; ld_xin = C; lens_t = w.lens2t;
  }
  {
    RowsNormArgs n;
    n.x_in = ht; n.ld_in = D; n.y = hnt_f; n.y_planes = hnt_p; n.ld_y = D; n.M = Mt; n.d = D; n.mode = NORM_ADA_RMS; n.eps = c.norm_eps; n.g1 = m.final_g;
    n.mod_a = mods + (size_t)(2 * depth) * 2 * D; n.mod_b = n.mod_a + D; n.ld_mod = 0; n.rows_per_batch = 0;
    if (rows_norm_forward(n, st)) return 1;
  }
  // long skip: skip_linear(cat[x_res, x]) (diffusion_transformer.py:243-244); x rows live in x_in[:, :C]
  if (gemm(m.skiplin_a, w.hn, D, w.hmid, D, Mt, st, ACT_NONE, nullptr, 0, hnt_p)) return 1;
  if (gemm(m.skiplin_b, xin_t, ld_xin, w.xres, D, Mt, st, ACT_NONE, w.hmid, D, nullptr, chain_wn_t ? w.xres_p : nullptr)) return 1;
  if (gemm(m.conv1, w.xres, D, w.wn_x, Wh, Mt, st, ACT_NONE, nullptr, 0, chain_wn_t ? w.xres_p : nullptr, chain_wn_t ? w.wnx_p : nullptr)) return 1;
  // WaveNet (wavenet.py:138-166)
  const int L = c.wn_layers, k = c.wn_kernel;
  for (int l = 0; l < L; ++l) {
    WNLayer& W = m.wn[l];
    int dil = 1;
    for (int q = 0; q < l; ++q) dil *= c.wn_dilation_rate;
    LinearWeights in = W.in_gate;
    in.bias = w.wnb + ((size_t)step * L + l) * 2 * Wh;     // in_layer bias + g_l of this step, gate-packed
    GemmArgs g;
    g.x = chain_wn_t ? nullptr : w.wn_x; g.x_planes = chain_wn_t ? w.wnx_p : nullptr; g.ldx = Wh; g.y = chain_t ? nullptr : w.wn_acts; g.y_planes = chain_t ? w.acts_p : nullptr; g.ldy = Wh; g.M = Mt; g.act = ACT_GATE;
    g.taps = k; g.seq_len = Tt; g.dil = dil; g.pad_left = (k - 1) / 2 * dil; g.pad_mode = 1; g.row_len = lens_t;
    if (gemm_forward(in, g, st)) return 1;
    if (W.has_res) {   // x = (x + res) * mask
      GemmArgs r;
      r.x = chain_t ? nullptr : w.wn_acts; r.x_planes = chain_t ? w.acts_p : nullptr; r.ldx = Wh; r.y = w.wn_x; r.y_planes = chain_wn_t ? w.wnx_p : nullptr; r.ldy = Wh; r.res = w.wn_x; r.ldr = Wh; r.M = Mt; r.seq_len = Tt; r.row_len = lens_t;
      if (gemm_forward(W.res, r, st)) return 1;
    }
    GemmArgs sk;       // output += skip ; the last layer's epilogue applies "* x_mask" to the finished sum
    sk.x = chain_t ? nullptr : w.wn_acts; sk.x_planes = chain_t ? w.acts_p : nullptr; sk.ldx = Wh; sk.y = w.wn_out; sk.ldy = Wh; sk.M = Mt;
    if (l > 0) { sk.res = w.wn_out; sk.ldr = Wh; }
    if (l == L - 1) { sk.seq_len = Tt; sk.row_len = lens_t; }
    if (gemm_forward(W.skip, sk, st)) return 1;
  }
  if (gemm(m.res_proj, w.xres, D, w.hmid, Wh, Mt, st, ACT_NONE, w.wn_out, Wh, chain_wn_t ? w.xres_p : nullptr)) return 1;
  {
    RowsNormArgs n;     // FinalLayer (diffusion_transformer.py:96-101)
    n.x_in = w.hmid; n.ld_in = Wh; n.y = hnt_f; n.y_planes = hnt_p; n.ld_y = Wh; n.M = Mt; n.d = Wh; n.mode = NORM_MOD_LN; n.eps = 1e-6f;
    n.mod_a = w.fmod + (size_t)step * 2 * Wh; n.mod_b = n.mod_a + Wh; n.ld_mod = 0; n.rows_per_batch = 0;
    if (rows_norm_forward(n, st)) return 1;
  }
  if (gemm(m.final_lin, w.hn, Wh, w.att, Wh, Mt, st, ACT_NONE, nullptr, 0, hnt_p)) return 1;
  return gemm(m.conv2, w.att, Wh, w.vout, C, Mt, st);
}

// ---- the two CFG halves on two streams -------------------------------------------------------------------
// Every kernel of the estimator alternates an MFMA-bound phase with an HBM-bound one (a GEMM tile: 57 k cycles of main loop, then
// 22-39 k cycles in which all 256 CUs write their tiles at once at ~4.6 TB/s; in-kernel stamps, profiles/README.md "Round 3"), and
// all workgroups of one launch are in the same phase.  Two launches of DIFFERENT kernels side by side are not: with the null
// half of the batch a couple of kernels behind the conditional half, one half's epilogues drain while the other half multiplies
// (tools/two_stream_gemm.py: the four GEMMs of a DiT layer 1290 -> 1098 us).  The halves share nothing between cfm_pack and the
// Euler update; results are bit-identical to the single-stream order (same kernels on the same rows).
struct HalfStreams { hipStream_t side = nullptr; hipEvent_t fork = nullptr, lag = nullptr, join = nullptr; };
static std::map<std::pair<int, hipStream_t>, HalfStreams> g_half_streams;   // per (device, caller stream)
static std::mutex g_half_mu;
// Off by default: alone on the device the solver takes 9 % less time with it (542 -> 491 ms at configs[2]), a sequential
// synthesize_batch 3 % less -- but in the serving pipeline the decode chains of the NEXT batches then run at less than half their
// speed (two of three lanes 2.65 s instead of 1.18 s per 512 tokens, also when decode and acoustic stages take strict turns; not
// understood: profiles/README.md "Round 3"), so the batch pipeline leaves it off.
static int g_s2mel_overlap = 0;
void set_s2mel_overlap(int on) { g_s2mel_overlap = on ? 1 : 0; }
int get_s2mel_overlap() { return g_s2mel_overlap; }

static int half_streams_for(hipStream_t st, HalfStreams* out) {
  int dev = 0;
  IDX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_half_mu);
  HalfStreams& hs = g_half_streams[std::make_pair(dev, st)];
  if (!hs.side) {
    // same priority as the caller's stream: the halves are peers (a side stream of HIGHER priority than its caller starves the
    // caller's half -- measured: the solver took twice as long on a low-priority serving stream)
    int prio = 0;
    if (st) IDX_HIP(hipStreamGetPriority(st, &prio));
    IDX_HIP(hipStreamCreateWithPriority(&hs.side, hipStreamNonBlocking, prio));
    IDX_HIP(hipEventCreateWithFlags(&hs.fork, hipEventDisableTiming));
    IDX_HIP(hipEventCreateWithFlags(&hs.lag, hipEventDisableTiming));
    IDX_HIP(hipEventCreateWithFlags(&hs.join, hipEventDisableTiming));
  }
  *out = hs;
  return 0;
}

int s2mel_release_stream(hipStream_t st) {      // idxtts_release_stream: the side stream + events kept for a caller stream
  int dev = 0;
  IDX_HIP(hipGetDevice(&dev));
  HalfStreams hs;
  {
    std::lock_guard<std::mutex> lk(g_half_mu);
    auto it = g_half_streams.find(std::make_pair(dev, st));
    if (it == g_half_streams.end()) return 0;
    hs = it->second;
    g_half_streams.erase(it);
  }
  if (hs.side) { (void)hipStreamSynchronize(hs.side); (void)hipStreamDestroy(hs.side); }
  if (hs.fork) (void)hipEventDestroy(hs.fork);
  if (hs.lag) (void)hipEventDestroy(hs.lag);
  if (hs.join) (void)hipEventDestroy(hs.join);
  return 0;
}

// true: the two halves run as two chains on two streams (half_view buffers); false: ONE stacked evaluation of 2B sequences on the
// caller's stream (larger launches: 394 instead of 2 x 198 tiles for an N = 512 GEMM fill the second round of CUs better)
static bool halves_on_two_streams(hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  return g_s2mel_overlap && !prof_enabled() && cap == hipStreamCaptureStatusNone;     // (per-launch event timing wants kernels alone)
}

// one estimator evaluation of the solver: both CFG halves, on two streams (two_streams) or stacked
static int dit_eval_halves(S2MelModel& m, CfmBuffers& w, int B, int T, int step, bool two_streams, hipStream_t st) {
  if (!two_streams) return dit_eval(m, w, 2 * B, T, step, st, nullptr);
  CfmBuffers v0 = half_view(m, w, 0, B, T);
  CfmBuffers v1 = half_view(m, w, 1, B, T);
  HalfStreams hs;
  if (half_streams_for(st, &hs)) return 1;
  IDX_HIP(hipEventRecord(hs.fork, st));
  IDX_HIP(hipStreamWaitEvent(hs.side, hs.fork, 0));
  // the null half starts when the conditional half has finished its first attention (a few kernels in): from then on the two
  // chains stay that far apart
  if (dit_eval(m, v0, B, T, step, st, hs.lag)) return 1;
  IDX_HIP(hipStreamWaitEvent(hs.side, hs.lag, 0));
  if (dit_eval(m, v1, B, T, step, hs.side, nullptr)) return 1;
  IDX_HIP(hipEventRecord(hs.join, hs.side));
  IDX_HIP(hipStreamWaitEvent(st, hs.join, 0));
  return 0;
}

int S2MelModel::cfm(const float* mu, const int* x_lens_host, const float* prompt, const int* prompt_lens_host, int Tp_max,
                    const float* style, const float* z, const float* t_emb, const float* dt_host, int n_steps, float cfg_rate,
                    float* out, int B, int T, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(mu && x_lens_host && prompt && prompt_lens_host && style && z && t_emb && dt_host && out, "null pointer");
  IDX_CHECK(B > 0 && T > 0 && n_steps > 0 && Tp_max > 0, "shape");
  IDX_CHECK(cfg_rate > 0.0f, "the stacked-CFG path needs inference_cfg_rate > 0 (reference default 0.7)");
  IDX_CHECK(T <= rope_len, "sequence longer than the rope cache");
  IDX_CHECK(ws && ws_bytes >= cfm_workspace_bytes(B, T, n_steps), "workspace too small");
  const int D = cfg.hidden_dim, C = cfg.in_channels, Wh = cfg.wn_hidden, depth = cfg.depth, L = cfg.wn_layers;
  CfmBuffers w = carve_cfm(*this, ws, B, T, n_steps);
  std::vector<int> lens2(2 * B), plen(B);
  for (int b = 0; b < B; ++b) {
    IDX_CHECK(x_lens_host[b] > 0 && x_lens_host[b] <= T && prompt_lens_host[b] >= 0 && prompt_lens_host[b] <= std::min(Tp_max, x_lens_host[b]), "lengths");
    lens2[b] = lens2[B + b] = x_lens_host[b];
    plen[b] = prompt_lens_host[b];
  }
  {
    // the post-transformer part runs on the frames from tail_t0 on: the shortest prompt minus the WaveNet's one-sided context
    int halo = 0, dil = 1, pmin = plen[0];
    for (int l = 0; l < L; ++l) { halo += (cfg.wn_kernel - 1) / 2 * dil; dil *= cfg.wn_dilation_rate; }
    for (int b = 0; b < B; ++b) pmin = std::min(pmin, plen[b]);
    w.tail_t0 = pmin - halo >= 64 ? pmin - halo : 0;
  }
  std::vector<int> lens2t(2 * B);
  for (int i = 0; i < 2 * B; ++i) lens2t[i] = lens2[i] - w.tail_t0;
  IDX_HIP(hipMemcpyAsync(w.lens2, lens2.data(), 2 * B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.lens2t, lens2t.data(), 2 * B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.plen, plen.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));
  // ---- per-call constants: every t-dependent vector for all steps at once (M = n_steps GEMMs) ----
  if (gemm(temb0, t_emb, 256, w.tmp_steps, D, n_steps, st, ACT_SILU) || gemm(temb2, w.tmp_steps, D, w.t1, D, n_steps, st)) return 1;
  if (gemm(mod_all, w.t1, D, w.mods, (2 * depth + 1) * 2 * D, n_steps, st)) return 1;
  if (silu_rows(w.t1s, w.t1, (size_t)n_steps * D, st) || gemm(final_mod, w.t1s, Wh, w.fmod, 2 * Wh, n_steps, st)) return 1;
  if (gemm(t2emb0, t_emb, 256, w.tmp_steps, Wh, n_steps, st, ACT_SILU) || gemm(t2emb2, w.tmp_steps, Wh, w.t2, Wh, n_steps, st)) return 1;
  if (gemm(wn_cond, w.t2, Wh, w.wnb, L * 2 * Wh, n_steps, st)) return 1;
  if (gemm(cond_proj, mu, cfg.content_dim, w.condp, D, B * T, st)) return 1;
  if (cfm_init_state(w.xstate, z, w.plen, B, C, T, st)) return 1;       // x[..., :prompt_len] = 0 (flow_matching.py:82)
  const bool two_streams = halves_on_two_streams(st);
  for (int s = 0; s < n_steps; ++s) {
    CfmPackArgs pk;
    pk.x_in = w.x_in; pk.ld = 2 * C + D + cfg.style_dim; pk.x = w.xstate; pk.prompt = prompt; pk.prompt_len = w.plen;
    pk.cond = w.condp; pk.cond_null = cond_proj.bias; pk.style = style;
    pk.B = B; pk.T = T; pk.C = C; pk.D = D; pk.S = cfg.style_dim; pk.Tp_max = Tp_max;
    pk.x_only = s > 0;      // x_in is read by the merge GEMM and the long skip only: its other 784 columns are packed once
    if (cfm_pack(pk, st)) return 1;
    if (dit_eval_halves(*this, w, B, T, s, two_streams, st)) return 1;
    CfmEulerArgs eu;      // (two streams: the null half's estimate lives in half_view's second block)
    eu.x = w.xstate; eu.v = w.vout; eu.v_null = two_streams ? w.vout + (size_t)B * T * C : nullptr; eu.ldv = C; eu.prompt_len = w.plen; eu.B = B; eu.T = T; eu.C = C; eu.dt = dt_host[s]; eu.cfg_rate = cfg_rate;
    eu.v_t0 = w.tail_t0; eu.v_T = T - w.tail_t0;
    if (cfm_euler(eu, st)) return 1;
  }
  IDX_HIP(hipMemcpyAsync(out, w.xstate, (size_t)B * C * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  return 0;
}

// One evaluation of the CFM estimator = DiT.forward(x, prompt_x, x_lens, t, style, cond) (diffusion_transformer.py:186-257)
// on B rows: the conditional half of the stacked [cond | null] evaluation the Euler loop runs every step.
int S2MelModel::estimator(const float* x, const float* prompt, const int* prompt_lens_host, int Tp_max, const int* x_lens_host,
                          const float* t_emb, const float* style, const float* mu, float* out_tm, int B, int T, void* ws, size_t ws_bytes,
                          hipStream_t st) {
  IDX_CHECK(x && prompt && prompt_lens_host && x_lens_host && t_emb && style && mu && out_tm, "null pointer");
  IDX_CHECK(B > 0 && T > 0 && Tp_max > 0 && T <= rope_len, "shape");
  IDX_CHECK(ws && ws_bytes >= cfm_workspace_bytes(B, T, 1), "workspace too small");
  const int D = cfg.hidden_dim, C = cfg.in_channels, Wh = cfg.wn_hidden, depth = cfg.depth, L = cfg.wn_layers;
  CfmBuffers w = carve_cfm(*this, ws, B, T, 1);
  std::vector<int> lens2(2 * B), plen(B);
  for (int b = 0; b < B; ++b) {
    IDX_CHECK(x_lens_host[b] > 0 && x_lens_host[b] <= T && prompt_lens_host[b] >= 0 && prompt_lens_host[b] <= std::min(Tp_max, T), "lengths");
    lens2[b] = lens2[B + b] = x_lens_host[b];
    plen[b] = prompt_lens_host[b];
  }
  IDX_HIP(hipMemcpyAsync(w.lens2, lens2.data(), 2 * B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.plen, plen.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));
  if (gemm(temb0, t_emb, 256, w.tmp_steps, D, 1, st, ACT_SILU) || gemm(temb2, w.tmp_steps, D, w.t1, D, 1, st)) return 1;
  if (gemm(mod_all, w.t1, D, w.mods, (2 * depth + 1) * 2 * D, 1, st)) return 1;
  if (silu_rows(w.t1s, w.t1, (size_t)D, st) || gemm(final_mod, w.t1s, Wh, w.fmod, 2 * Wh, 1, st)) return 1;
  if (gemm(t2emb0, t_emb, 256, w.tmp_steps, Wh, 1, st, ACT_SILU) || gemm(t2emb2, w.tmp_steps, Wh, w.t2, Wh, 1, st)) return 1;
  if (gemm(wn_cond, w.t2, Wh, w.wnb, L * 2 * Wh, 1, st)) return 1;
  if (gemm(cond_proj, mu, cfg.content_dim, w.condp, D, B * T, st)) return 1;
  IDX_HIP(hipMemcpyAsync(w.xstate, x, (size_t)B * C * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  CfmPackArgs pk;
  pk.x_in = w.x_in; pk.ld = 2 * C + D + cfg.style_dim; pk.x = w.xstate; pk.prompt = prompt; pk.prompt_len = w.plen;
  pk.cond = w.condp; pk.cond_null = cond_proj.bias; pk.style = style;
  pk.B = B; pk.T = T; pk.C = C; pk.D = D; pk.S = cfg.style_dim; pk.Tp_max = Tp_max;
  if (cfm_pack(pk, st)) return 1;
  {
    CfmBuffers v0 = half_view(*this, w, 0, B, T);      // the conditional half alone
    if (dit_eval(*this, v0, B, T, 0, st, nullptr)) return 1;
  }
  IDX_HIP(hipMemcpyAsync(out_tm, w.vout, (size_t)B * T * C * sizeof(float), hipMemcpyDeviceToDevice, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------

static CondBuffers carve_cond(const S2MelModel& m, void* ws, int B, int M, int Tg) {
  const auto& c = m.cfg;
  CondBuffers b;
  Carver2 k(ws);
  const size_t rows = (size_t)B * std::max(M, Tg);
  const int wide = std::max(std::max(c.codec_hidden, c.lr_channels), std::max(c.gpt_layer_dims[0], c.gpt_layer_dims[1]));
  b.a = k.take<float>(rows * wide);
  b.b = k.take<float>(rows * wide);
  b.s = k.take<float>(rows * wide);
  b.stats = k.take<float>(2 * B);
  b.idx_code = k.take<int>((size_t)B * M);
  b.idx_row = k.take<int>((size_t)B * M);
  b.idx_interp = k.take<int>((size_t)B * Tg);
  b.tlen = k.take<int>(B);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

size_t S2MelModel::cond_workspace_bytes(int B, int M, int Tg) const { return carve_cond(*this, nullptr, B, M, Tg).bytes; }

int S2MelModel::prepare_cond(const float* latent, const long long* codes, const int* code_lens_host, const int* target_lens_host,
                             int B, int M, int Tg, float* cond_out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(latent && codes && code_lens_host && target_lens_host && cond_out, "null pointer");
  IDX_CHECK(B > 0 && M > 0 && Tg > 0, "shape");
  IDX_CHECK(ws && ws_bytes >= cond_workspace_bytes(B, M, Tg), "workspace too small");
  const int Hc = cfg.codec_hidden;
  CondBuffers w = carve_cond(*this, ws, B, M, Tg);
  // host-side index tables: code ids, and torch's 'nearest' source row for every target frame
  std::vector<long long> hc((size_t)B * M);
  IDX_HIP(hipMemcpyAsync(hc.data(), codes, hc.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipStreamSynchronize(st));
  std::vector<int> ic((size_t)B * M), ir((size_t)B * M), ii((size_t)B * Tg, -1), tl(B);
  for (int b = 0; b < B; ++b) {
    const int mb = code_lens_host[b], tb = target_lens_host[b];
    IDX_CHECK(mb > 0 && mb <= M && tb > 0 && tb <= Tg, "code_lens / target_lens out of range");
    tl[b] = tb;
    for (int i = 0; i < M; ++i) {
      const long long cde = hc[(size_t)b * M + i];
      IDX_CHECK(i >= mb || (cde >= 0 && cde < cfg.codebook_size), "semantic code out of the codebook");
      ic[(size_t)b * M + i] = i < mb ? (int)cde : -1;
      ir[(size_t)b * M + i] = i < mb ? b * M + i : -1;
    }
    // F.interpolate(mode='nearest'): src = min(floor(dst * (float)in / out), in - 1), float32 arithmetic
    const float scale = (float)mb / (float)tb;
    for (int j = 0; j < tb; ++j) ii[(size_t)b * Tg + j] = b * M + std::min((int)std::floor((float)j * scale), mb - 1);
  }
  IDX_HIP(hipMemcpyAsync(w.idx_code, ic.data(), ic.size() * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.idx_row, ir.data(), ir.size() * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.idx_interp, ii.data(), ii.size() * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.tlen, tl.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));
  const int rowsM = B * M;
  // gpt_layer: 1280 -> 256 -> 128 -> 1024 (commons.py:413)
  if (gemm(gl[0], latent, cfg.gpt_dim, w.a, gl[0].N, rowsM, st) || gemm(gl[1], w.a, gl[0].N, w.b, gl[1].N, rowsM, st) ||
      gemm(gl[2], w.b, gl[1].N, w.a, Hc, rowsM, st)) return 1;
  // S_infer = vq2emb(codes) + latent (infer_v2.py:841-843)
  GatherArgs ga;
  ga.out = w.s; ga.ld_out = Hc; ga.d = Hc;
  ga.table[0] = vq_table; ga.idx[0] = w.idx_code;
  ga.table[1] = w.a; ga.idx[1] = w.idx_row;
  if (gather_sum_rows(ga, rowsM, st)) return 1;
  return regulate_rows(w.s, w, B, M, Tg, cond_out, st);
}

// InterpolateRegulator.forward on token-major rows s [B*M][lr_in_channels] (length_regulator.py:117-141): content_in_proj, nearest
// interpolation M_b -> Tg_b through the host-built index table w.idx_interp, 4 x (Conv1d k3 -> GroupNorm(1) -> Mish), Conv1d 1x1, mask
int S2MelModel::regulate_rows(const float* s_rows, const CondBuffers& w, int B, int M, int Tg, float* cond_out, hipStream_t st) {
  const int Hc = cfg.codec_hidden, LC = cfg.lr_channels;
  const int rowsM = B * M, rowsT = B * Tg;
  // content_in_proj, nearest interpolation M_b -> Tg_b (rows beyond Tg_b are zero)
  if (gemm(lr_in, s_rows, Hc, w.a, LC, rowsM, st)) return 1;
  GatherArgs gi;
  gi.out = w.b; gi.ld_out = LC; gi.d = LC; gi.table[0] = w.a; gi.idx[0] = w.idx_interp;
  if (gather_sum_rows(gi, rowsT, st)) return 1;
  float* cur = w.b;
  float* other = w.a;
  for (int n = 0; n < cfg.lr_num_convs; ++n) {
    GemmArgs g;
    g.x = cur; g.ldx = LC; g.y = other; g.ldy = LC; g.M = rowsT; g.taps = 3; g.seq_len = Tg; g.dil = 1; g.pad_left = 1; g.pad_mode = 0;
    if (gemm_forward(lr_conv[n], g, st)) return 1;
    if (groupnorm1_mish(cur, other, lr_gn_g[n], lr_gn_b[n], w.tlen, B, Tg, LC, 1e-5f, w.stats, st)) return 1;
  }
  GemmArgs o;
  o.x = cur; o.ldx = LC; o.y = cond_out; o.ldy = LC; o.M = rowsT; o.seq_len = Tg; o.row_len = w.tlen;    // * mask
  return gemm_forward(lr_out, o, st);
}

// length_regulator(S, ylens) alone = the prompt-side call of infer_v2.py:649-652 (S_ref -> prompt_condition)
int S2MelModel::regulate(const float* S, const int* in_lens_host, const int* target_lens_host, int B, int M, int Tg, float* cond_out,
                         void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(S && in_lens_host && target_lens_host && cond_out, "null pointer");
  IDX_CHECK(B > 0 && M > 0 && Tg > 0, "shape");
  IDX_CHECK(ws && ws_bytes >= cond_workspace_bytes(B, M, Tg), "workspace too small");
  CondBuffers w = carve_cond(*this, ws, B, M, Tg);
  std::vector<int> ii((size_t)B * Tg, -1), tl(B);
  for (int b = 0; b < B; ++b) {
    const int mb = in_lens_host[b], tb = target_lens_host[b];
    IDX_CHECK(mb > 0 && mb <= M && tb > 0 && tb <= Tg, "in_lens / target_lens out of range");
    tl[b] = tb;
    const float scale = (float)mb / (float)tb;      // F.interpolate(mode='nearest'), float32 arithmetic
    for (int j = 0; j < tb; ++j) ii[(size_t)b * Tg + j] = b * M + std::min((int)std::floor((float)j * scale), mb - 1);
  }
  IDX_HIP(hipMemcpyAsync(w.idx_interp, ii.data(), ii.size() * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemcpyAsync(w.tlen, tl.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));
  return regulate_rows(S, w, B, M, Tg, cond_out, st);
}

}  // namespace idxtts
