// IndexTTS-2 GPT stage on MI355X: prefill, KV-cached greedy decode and the latent pass.
//
// Reference: UnifiedVoice.inference_speech model_v2.py:796-895 (decode through GPT2InferenceModel.forward
// 131-225 + GenerationMixin._sample, or the accel_engine.generate plugin slot 871-883), and
// UnifiedVoice.forward 673-723 (latent pass).  The 24 GPT-2 blocks are third-party HF code
// (transformers_gpt2.py:615-674 is the in-tree spec).
//
// Two weight copies are kept on purpose (288 GB HBM): an MFMA-32x32 packed copy for the GEMM-shaped
// passes (prefill, latent: M = B*S rows) and a B-fragment stream-order copy for the M<=64 decode GEMVs.
//
// Decode step = 5 launches per layer, every activation an MFMA A-fragment image (gemv_fx.hip):
//   c_attn [LayerNorm 1 folded in] -> decode_attn (+ KV-cache write) -> c_proj (+ residual, in place)
//   -> c_fc [LayerNorm 2 folded in, gelu_new] -> mlp.c_proj (+ residual; K split over 4 workgroups per column tile, finished
//   by the last one to arrive),
// + final norms, head and ONE launch for sampler + next embedding + advance (greedy): 123 launches per token.  Every per-step scalar is device-resident, so the whole
// step is captured once into a hipGraph and replayed per token (launch-bound otherwise).
// The decode weight streams are fp32 by default; quantize_weights() rounds the model once to bf16 / fp8-e4m3 storage
// (BASELINE configs[4]) -- the arithmetic stays the fp32 MFMA.
#include <cstdlib>
#include <cstring>

#include "gpt.h"

namespace idxtts {

GPTModel::GPTModel(const idxtts_gpt_config& c) : cfg(c) {}

bool GPTModel::accepts(const std::string& name) const {
  static const char* prefixes[] = {"gpt.h.", "gpt.ln_f.", "final_norm.", "mel_head.", "mel_embedding.", "text_embedding.",
                                   "mel_pos_embedding.", "text_pos_embedding.", "speed_emb."};
  for (const char* p : prefixes)
    if (name.rfind(p, 0) == 0) return true;
  return false;
}

static int need(std::map<std::string, HostTensor>& t, const std::string& key, std::vector<int64_t> shape, HostTensor** out) {
  auto it = t.find(key);
  if (it == t.end()) IDX_FAIL("missing tensor '" + key + "'");
  if (it->second.shape != shape) IDX_FAIL("tensor '" + key + "' has the wrong shape");
  *out = &it->second;
  return 0;
}

static int upload(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& key, std::vector<int64_t> shape,
                  const float** out) {
  HostTensor* v = nullptr;
  if (need(t, key, shape, &v)) return 1;
  float* d = nullptr;
  if (arena.upload(v->data.data(), v->data.size(), &d)) return 1;
  *out = d;
  return 0;
}

// HF Conv1D weight [K][N] -> both packed forms
// ln_g / ln_b (host, [K]) non-null: the decode copy gets the LayerNorm in front of it folded in (gemv_fx.hip):
//   W' = diag(g) W (packed), u = colsum(W'), c = b . W + bias
// decode stream in the model's storage format (the fp32 pack `buf` holds values the format represents exactly)
static int upload_stream(DeviceArena& arena, const std::vector<float>& buf, int N, int K, int fmt, Gemv16Weights* gw) {
  gw->N = N; gw->K = K; gw->fmt = fmt;
  if (fmt == WFMT_F32) {
    float* d = nullptr;
    if (arena.upload(buf.data(), buf.size(), &d)) return 1;
    gw->wp = d;
    return 0;
  }
  std::vector<unsigned char> cbuf(buf.size() * wfmt_bytes(fmt));
  std::vector<float> scale(N, 1.0f);
  if (compact_gemv16(cbuf.data(), buf.data(), N, K, fmt, scale.data())) IDX_FAIL("decode weights are not representable in the compact format (quantize_weights not applied?)");
  void* d = nullptr;
  if (arena.upload_bytes(cbuf.data(), cbuf.size(), &d)) return 1;
  gw->wp = d;
  if (fmt == WFMT_FP8) {
    float* ds = nullptr;
    if (arena.upload(scale.data(), scale.size(), &ds)) return 1;
    gw->wscale = ds;
  }
  return 0;
}

// the same matrix in the plane GEMV's order (compact formats only; gemv_pl.h)
static int upload_stream32(DeviceArena& arena, const float* w, int N, int K, bool kn, int fmt, Gemv32Weights* gp) {
  gp->N = N; gp->K = K; gp->fmt = fmt;
  if (fmt == WFMT_F32) return 0;
  std::vector<unsigned char> cbuf(gemv32_packed_elems(N, K) * wfmt_bytes(fmt));
  std::vector<float> scale(N, 1.0f);
  if (pack_gemv32(cbuf.data(), w, N, K, kn, fmt, scale.data())) IDX_FAIL("decode weights are not representable in the compact format (quantize_weights not applied?)");
  void* d = nullptr;
  if (arena.upload_bytes(cbuf.data(), cbuf.size(), &d)) return 1;
  gp->wp = d;
  if (fmt == WFMT_FP8) {
    float* ds = nullptr;
    if (arena.upload(scale.data(), scale.size(), &ds)) return 1;
    gp->wscale = ds;
  }
  return 0;
}

static int make_proj(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& prefix, int K, int N, int fmt,
                     LinearWeights* lw, Gemv16Weights* gw, Gemv32Weights* gp, const HostTensor* ln_g = nullptr, const HostTensor* ln_b = nullptr,
                     const float** u_out = nullptr, const float** c_out = nullptr) {
  HostTensor* w = nullptr;
  if (need(t, prefix + ".weight", {K, N}, &w)) return 1;
  std::vector<float> buf(linear_packed_floats(N, K));
  pack_linear_kn(buf.data(), w->data.data(), K, N);
  float* d = nullptr;
  if (arena.upload(buf.data(), buf.size(), &d)) return 1;
  lw->wp = d; lw->N = N; lw->K = K;
  if (upload(t, arena, prefix + ".bias", {N}, &lw->bias)) return 1;
  {   // split-bf16 copy, used by the latent pass only (the KV-cache-building prefill stays exact fp32)
    std::vector<float> wt((size_t)N * K);
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) wt[(size_t)n * K + k] = w->data[(size_t)k * N + n];
    std::vector<float> p16((linear_bf16x3_packed_bytes(N, K) + 3) / 4);
    pack_linear_bf16x3(p16.data(), wt.data(), N, K);
    float* d16 = nullptr;
    if (arena.upload(p16.data(), p16.size(), &d16)) return 1;
    lw->wp16 = d16;
  }
  buf.assign(gemv16_packed_floats(N, K), 0.0f);
  if (ln_g) {
    HostTensor* bias = nullptr;
    if (need(t, prefix + ".bias", {N}, &bias)) return 1;
    std::vector<float> wf((size_t)K * N), u(N), c(N);
    std::vector<double> ud(N, 0.0), cd(N, 0.0);
    for (int k = 0; k < K; ++k) {
      const float g = ln_g->data[k];
      const double b = ln_b->data[k];
      for (int n = 0; n < N; ++n) {
        const float wv = w->data[(size_t)k * N + n];
        const float wg = g * wv;
        wf[(size_t)k * N + n] = wg;
        ud[n] += (double)wg;
        cd[n] += b * (double)wv;
      }
    }
    for (int n = 0; n < N; ++n) { u[n] = (float)ud[n]; c[n] = (float)(cd[n] + (double)bias->data[n]); }
    pack_gemv16_kn(buf.data(), wf.data(), K, N);
    float *du = nullptr, *dc = nullptr;
    if (arena.upload(u.data(), u.size(), &du) || arena.upload(c.data(), c.size(), &dc)) return 1;
    *u_out = du; *c_out = dc;
    if (gp && upload_stream32(arena, wf.data(), N, K, true, fmt, gp)) return 1;
  } else {
    pack_gemv16_kn(buf.data(), w->data.data(), K, N);
    if (gp && upload_stream32(arena, w->data.data(), N, K, true, fmt, gp)) return 1;
  }
  return upload_stream(arena, buf, N, K, fmt, gw);
}

int GPTModel::quantize_weights(std::map<std::string, HostTensor>& t, int fmt) {
  IDX_CHECK(fmt == WFMT_F32 || fmt == WFMT_BF16 || fmt == WFMT_FP8, "weight format: 0 fp32, 1 bf16, 2 fp8-e4m3 + power-of-two column scale");
  IDX_CHECK(weight_fmt == WFMT_F32, "weights already quantised");
  if (fmt == WFMT_F32) return 0;
  if (fmt == WFMT_FP8 && fp8_check_device_decode(nullptr)) return 1;
  const int d = cfg.model_dim, f = 4 * d;
  auto fold = [&](const std::string& ln, const std::string& proj, int K, int N) -> int {
    HostTensor *g = nullptr, *b = nullptr, *w = nullptr, *bias = nullptr;
    if (need(t, ln + ".weight", {K}, &g) || need(t, ln + ".bias", {K}, &b) || need(t, proj + ".weight", {K, N}, &w) ||
        need(t, proj + ".bias", {N}, &bias)) return 1;
    std::vector<double> cd(N, 0.0);
    for (int k = 0; k < K; ++k) {
      const float gk = g->data[k];
      const double bk = b->data[k];
      float* row = &w->data[(size_t)k * N];
      for (int n = 0; n < N; ++n) { cd[n] += bk * (double)row[n]; row[n] = gk * row[n]; }
    }
    for (int n = 0; n < N; ++n) bias->data[n] = (float)(cd[n] + (double)bias->data[n]);
    quantize_matrix(w->data.data(), K, N, true, fmt);
    std::fill(g->data.begin(), g->data.end(), 1.0f);
    std::fill(b->data.begin(), b->data.end(), 0.0f);
    return 0;
  };
  auto plain = [&](const std::string& proj, int K, int N) -> int {
    HostTensor* w = nullptr;
    if (need(t, proj + ".weight", {K, N}, &w)) return 1;
    quantize_matrix(w->data.data(), K, N, true, fmt);
    return 0;
  };
  for (int i = 0; i < cfg.layers; ++i) {
    const std::string p = "gpt.h." + std::to_string(i);
    if (fold(p + ".ln_1", p + ".attn.c_attn", d, 3 * d) || plain(p + ".attn.c_proj", d, d)) return 1;
    if (fold(p + ".ln_2", p + ".mlp.c_fc", d, f) || plain(p + ".mlp.c_proj", f, d)) return 1;
  }
  HostTensor* hw = nullptr;
  if (need(t, "mel_head.weight", {cfg.number_mel_codes, d}, &hw)) return 1;
  quantize_matrix(hw->data.data(), d, cfg.number_mel_codes, false, fmt);
  weight_fmt = fmt;
  return 0;
}

int GPTModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int d = cfg.model_dim, f = 4 * d;
  IDX_CHECK(d == cfg.heads * 64, "head_dim must be 64");
  IDX_CHECK(cfg.layers > 0 && (d & 15) == 0, "config");
  {
    const int zeros[OOB_SLOTS] = {0};
    void* flag = nullptr;
    if (arena.upload_bytes(zeros, sizeof(zeros), &flag)) return 1;
    oob_flag = static_cast<int*>(flag);
  }
  layers.resize(cfg.layers);
  for (int i = 0; i < cfg.layers; ++i) {
    GPTLayer& L = layers[i];
    const std::string p = "gpt.h." + std::to_string(i);
    if (upload(t, arena, p + ".ln_1.weight", {d}, &L.ln1_g) || upload(t, arena, p + ".ln_1.bias", {d}, &L.ln1_b)) return 1;
    if (upload(t, arena, p + ".ln_2.weight", {d}, &L.ln2_g) || upload(t, arena, p + ".ln_2.bias", {d}, &L.ln2_b)) return 1;
    HostTensor *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    if (need(t, p + ".ln_1.weight", {d}, &g1) || need(t, p + ".ln_1.bias", {d}, &b1)) return 1;
    if (need(t, p + ".ln_2.weight", {d}, &g2) || need(t, p + ".ln_2.bias", {d}, &b2)) return 1;
    Gemv32Weights* const no32 = nullptr;
    const bool p32 = weight_fmt != WFMT_F32 && d % 32 == 0;
    if (make_proj(t, arena, p + ".attn.c_attn", d, 3 * d, weight_fmt, &L.attn_l, &L.attn_g, p32 ? &L.attn_p : no32, g1, b1, &L.attn_u, &L.attn_c)) return 1;
    if (make_proj(t, arena, p + ".attn.c_proj", d, d, weight_fmt, &L.proj_l, &L.proj_g, p32 ? &L.proj_p : no32)) return 1;
    if (make_proj(t, arena, p + ".mlp.c_fc", d, f, weight_fmt, &L.fc_l, &L.fc_g, p32 ? &L.fc_p : no32, g2, b2, &L.fc_u, &L.fc_c)) return 1;
    if (make_proj(t, arena, p + ".mlp.c_proj", f, d, weight_fmt, &L.fc2_l, &L.fc2_g, p32 ? &L.fc2_p : no32)) return 1;
  }
  if (upload(t, arena, "gpt.ln_f.weight", {d}, &lnf_g) || upload(t, arena, "gpt.ln_f.bias", {d}, &lnf_b)) return 1;
  if (upload(t, arena, "final_norm.weight", {d}, &fn_g) || upload(t, arena, "final_norm.bias", {d}, &fn_b)) return 1;
  const int V = cfg.number_mel_codes;
  HostTensor* hw = nullptr;
  if (need(t, "mel_head.weight", {V, d}, &hw)) return 1;
  std::vector<float> buf(gemv16_packed_floats(V, d));
  pack_gemv16_nk(buf.data(), hw->data.data(), V, d);
  if (upload_stream(arena, buf, V, d, weight_fmt, &head_g)) return 1;
  if (weight_fmt != WFMT_F32 && d % 32 == 0 && upload_stream32(arena, hw->data.data(), V, d, false, weight_fmt, &head_p)) return 1;
  if (upload(t, arena, "mel_head.bias", {V}, &head_b)) return 1;
  if (upload(t, arena, "mel_embedding.weight", {V, d}, &mel_emb)) return 1;
  if (upload(t, arena, "text_embedding.weight", {cfg.number_text_tokens + 1, d}, &text_emb)) return 1;
  if (upload(t, arena, "mel_pos_embedding.emb.weight", {cfg.mel_pos_len, d}, &mel_pos)) return 1;
  if (upload(t, arena, "text_pos_embedding.emb.weight", {cfg.text_pos_len, d}, &text_pos)) return 1;
  return 0;
}

// ---- workspace carving ----
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

GPTModel::Buffers GPTModel::carve(void* ws, int B, int S, int max_new) const {
  const int d = cfg.model_dim, V = cfg.number_mel_codes, L = cfg.layers;
  Buffers b;
  Carver c(ws);
  const size_t rows = (size_t)B * S;
  b.x = c.take<float>(rows * d);
  b.h = c.take<float>(rows * d);
  b.qkv = c.take<float>(rows * 3 * d);
  b.att = c.take<float>(rows * d);
  b.ff = c.take<float>(rows * 4 * d);
  b.Smax = max_new > 0 ? ((S + max_new + 3) & ~3) : 0;
  const size_t cache = (size_t)L * kv_layer_bytes(B, b.Smax);
  b.kcache = c.take<char>(cache);
  b.vcache = c.take<char>(cache);
  // decode activations as A-fragment images (frag_index, common.h); one contiguous region so it can be zeroed in one go
  b.xd = c.take<float>(frag_image_floats(B, d));
  b.frag_off = c.off - frag_image_floats(B, d) * sizeof(float);
  b.hd = c.take<float>(frag_image_floats(B, d));
  b.attd = c.take<float>(frag_image_floats(B, d));
  b.ffd = c.take<float>(frag_image_floats(B, 4 * d));
  b.pl_cnt = c.take<unsigned>((size_t)cdiv(std::max(V, 4 * d), 16));      // (arrival counters of the plane GEMV: part of the zeroed region)
  b.frag_bytes = c.off - b.frag_off;
  b.xrow = c.take<float>((size_t)B * d);
  b.hrow = c.take<float>((size_t)B * d);
  b.attrow = c.take<float>((size_t)B * d);
  b.ffrow = c.take<float>((size_t)B * 4 * d);
  b.stats = c.take<float>((size_t)cdiv(d, 16) * cdiv(B, 16) * 16 * 2);
  b.pl_slab = c.take<float>(std::max({gemv_pl_slab_floats(3 * d, d, B), gemv_pl_slab_floats(d, d, B), gemv_pl_slab_floats(4 * d, d, B),
                                      gemv_pl_slab_floats(d, 4 * d, B), gemv_pl_slab_floats(V, d, B), (size_t)64}));
  b.qkvd = c.take<float>((size_t)B * 3 * d);
  b.slab = c.take<float>((size_t)8 * B * d);
  b.logits = c.take<float>((size_t)B * V);
  b.seen = c.take<unsigned char>((size_t)B * V);
  b.finished = c.take<int>(B);
  b.cur_tok = c.take<int>(B);
  b.kstart = c.take<int>(B);
  b.ksb_cnt = c.take<unsigned>((size_t)cdiv(d, 16));
  b.attn_cnt = c.take<unsigned>((size_t)B * cfg.heads);
  b.attn_part = c.take<float>((size_t)B * cfg.heads * 16 * 66);
  b.state = c.take<DecodeState>(1);
  b.codes = c.take<long long>((size_t)B * (max_new > 0 ? max_new : 0));
  b.bytes = (c.off + 255) & ~(size_t)255;
  return b;
}

size_t GPTModel::workspace_bytes(int B, int S, int max_new) const { return carve(nullptr, B, S, max_new).bytes; }

// one transformer layer over M = B*S token rows (prefill / latent pass)
int GPTModel::layer_full(int li, const Buffers& w, int B, int S, const int* kstart, bool store_kv, hipStream_t st) {
  // The prefill (store_kv) that fills an fp32 KV cache feeds an exact greedy decode: exact fp32 MFMA.  With a bf16 cache its keys and
  // values are rounded to 8 bits on the way in (relative 2^-9), which buries the split-bf16 GEMM's 2^-16 product error: that prefill
  // runs like the latent pass, mode-dependent (split-bf16 by default, 3-4 x the exact kernel's rate).
  const bool exact = store_kv && kv_fmt == 0;
  auto mm = [&](const LinearWeights& lw, const GemmArgs& ga) { return exact ? gemm_tn_forward(lw, ga, st) : gemm_forward(lw, ga, st); };
  const GPTLayer& L = layers[li];
  const int d = cfg.model_dim, M = B * S;
  RowsNormArgs n1;
  n1.x_in = w.x; n1.ld_in = d; n1.y = w.h; n1.ld_y = d; n1.M = M; n1.d = d; n1.mode = NORM_LN; n1.g1 = L.ln1_g; n1.b1 = L.ln1_b;
  if (rows_norm_forward(n1, st)) return 1;
  GemmArgs g;
  g.x = w.h; g.ldx = d; g.y = w.qkv; g.ldy = 3 * d; g.M = M;
  if (mm(L.attn_l, g)) return 1;
  if (store_kv) {
    const size_t per_layer = kv_layer_bytes(B, w.Smax);
    if (kv_store_prefill(w.qkv, w.kcache + li * per_layer, w.vcache + li * per_layer, kv_fmt, B, cfg.heads, S, w.Smax, d, st)) return 1;
  }
  AttnArgs a;
  a.q = w.qkv; a.k = w.qkv + d; a.v = w.qkv + 2 * d; a.o = w.att;
  a.q_bs = a.k_bs = a.v_bs = (long)S * 3 * d; a.o_bs = (long)S * d;
  a.q_ts = a.k_ts = a.v_ts = 3 * d; a.o_ts = d;
  a.B = B; a.H = cfg.heads; a.Sq = S; a.Sk = S; a.causal = 1; a.kstart = kstart; a.scale = 0.125f;
  a.split_bf16 = !exact && get_gemm_mode() == GEMM_BF16X3;      // the prefill of an fp32 cache stays exact fp32
  if (flash_attn_forward(a, st)) return 1;
  GemmArgs p;
  p.x = w.att; p.ldx = d; p.y = w.x; p.ldy = d; p.res = w.x; p.ldr = d; p.M = M;
  if (mm(L.proj_l, p)) return 1;
  RowsNormArgs n2 = n1;
  n2.g1 = L.ln2_g; n2.b1 = L.ln2_b;
  if (rows_norm_forward(n2, st)) return 1;
  GemmArgs f1;
  f1.x = w.h; f1.ldx = d; f1.y = w.ff; f1.ldy = 4 * d; f1.M = M; f1.act = ACT_GELU_NEW;
  if (mm(L.fc_l, f1)) return 1;
  GemmArgs f2;
  f2.x = w.ff; f2.ldx = 4 * d; f2.y = w.x; f2.ldy = d; f2.res = w.x; f2.ldr = d; f2.M = M;
  if (mm(L.fc2_l, f2)) return 1;
  return 0;
}

// sampling mode of the generation this host thread is running (mode 0 = greedy)
static thread_local idxtts_sampling samp{0, 1.0f, 0, 1.0f, nullptr, 0};
thread_local const BeamState* tl_beam = nullptr;
thread_local int tl_prof_pos = 0;
static thread_local const long long* tl_forced = nullptr;      // teacher-forced generation in flight: [B][max_new] tokens fed back instead of the argmax
static thread_local int tl_forced_ld = 0;      // keys the eager decode step in flight reads (0 while a captured graph replays)

// head on B rows: ln_f -> final_norm (one rows_norm launch, output as fragment images) -> mel_head -> greedy sampler
int GPTModel::head_and_sample(const Buffers& w, int B, const float* x, int ldx, bool x_frag, float penalty, long long* codes,
                              int codes_ld, float* logits_out, hipStream_t st) {
  const int V = cfg.number_mel_codes, d = cfg.model_dim;
  const bool pl = use_pl(B);
  RowsNormArgs n;
  n.x_in = x; n.ld_in = ldx; n.in_frag = x_frag ? 1 : 0; n.M = B; n.d = d;
  if (pl) { n.y = w.hrow; n.ld_y = d; } else { n.y = w.hd; n.ld_y = d; n.y_frag = 1; }
  n.mode = NORM_LN_LN; n.g1 = lnf_g; n.b1 = lnf_b; n.g2 = fn_g; n.b2 = fn_b;
  if (rows_norm_forward(n, st)) return 1;
  if (pl) {
    GemvPLArgs hv;
    hv.x = w.hrow; hv.ldx = d; hv.rows = B; hv.bias = head_b; hv.y = w.logits; hv.ldy = V; hv.slab = w.pl_slab; hv.counters = w.pl_cnt;
    if (gemv_pl_forward(head_p, hv, st)) return 1;
  } else {
    GemvFXArgs hv;
    hv.xf = w.hd; hv.rows = B; hv.bias = head_b; hv.y = w.logits; hv.ldy = V;
    if (gemv_fx_forward(head_g, hv, st)) return 1;
  }
  if (tl_beam) return beam_scores_forward(*tl_beam, st) || beam_select_forward(*tl_beam, st) || beam_reorder_forward(*tl_beam, st);
  SampleArgs s;
  s.part = w.logits; s.parts = 1; s.part_rows = B; s.bias = nullptr; s.logits_out = logits_out;
  s.seen = w.seen; s.finished = w.finished; s.codes = codes; s.codes_ld = codes_ld; s.cur_tok = w.cur_tok;
  s.st = w.state; s.B = B; s.V = V; s.stop_token = cfg.stop_mel_token; s.penalty = penalty;
  s.forced = tl_forced; s.forced_ld = tl_forced_ld;
  if (fused_tail(B)) {      // the sampler's workgroups also write the next step's input and advance the step scalars
    if (pl) { s.embed.x_row = w.xrow; s.embed.x_stats = w.stats; } else s.embed.x_frag = w.xd;
    s.embed.mel_emb = mel_emb; s.embed.mel_pos = mel_pos; s.embed.d = d; s.embed.st_rw = w.state;
  }
  if (samp.mode != 0) {
    SampleWarpArgs sw;
    sw.base = s; sw.mode = samp.mode; sw.temperature = samp.temperature; sw.top_k = samp.top_k; sw.top_p = samp.top_p;
    sw.exp_noise = samp.exp_noise; sw.seed = samp.seed;
    return sample_warp_forward(sw, st);
  }
  return sample_greedy_forward(s, st);
}

// one autoregressive step for all B rows (replayable: no host-dependent arguments); 5 launches per layer, every
// activation a fragment image:  c_attn [LN1 folded] -> attention -> c_proj (+x, in place) -> c_fc [LN2 folded, gelu_new]
// -> mlp.c_proj (+x, in place)
// The decode step runs on the plane GEMV from idxtts_set_decode_plane_rows() rows on (default 17: two or more MFMA row tiles -- merged
// requests, 16 utterances x 3 beams).  Measured on the full-size model, 256 tokens, graph replay: 16 rows 0.328 s against 0.303 s on the
// fp32-MFMA GEMV (its single-round-trip launches win), 32 rows 0.432 / 0.432, 48 rows 0.567 / 0.612 (profiles/README.md "Round 4").
bool GPTModel::use_pl(int B) const { return weight_fmt != WFMT_F32 && B >= get_decode_plane_rows() && cfg.model_dim % 32 == 0 && head_p.wp; }
bool GPTModel::fused_tail(int B) const { (void)B; return samp.mode == 0 && !tl_beam; }      // greedy: sample + next embedding + advance are ONE launch

// The decode step on the plane GEMV (gemv_pl.hip): the same five launches per layer, every activation a plain fp32 row-major matrix
// (split into bf16 planes inside the GEMV), the residual stream updated in place, the LayerNorm statistics handed from producer to consumer; greedy generations close the
// step with ONE launch (sample + next embedding + advance).
int GPTModel::decode_step_pl(const Buffers& w, int B, float penalty, long long* codes, int codes_ld, float* logits_base, hipStream_t st) {
  const int d = cfg.model_dim, T = d / 16;
  const size_t per_layer = kv_layer_bytes(B, w.Smax);
  const bool fused = fused_tail(B);
  if (!fused && embed_step_pl(w.xrow, w.stats, B, d, mel_emb, mel_pos, w.cur_tok, w.state, st)) return 1;
  for (int li = 0; li < cfg.layers; ++li) {
    const GPTLayer& L = layers[li];
    GemvPLArgs qa;      // qkv = c_attn(LN1(x)) + b, row-major for the attention kernel
    qa.x = w.xrow; qa.ldx = d; qa.rows = B; qa.colsum = L.attn_u; qa.bias = L.attn_c; qa.stats_in = w.stats; qa.stats_tiles = T;
    qa.y = w.qkvd; qa.ldy = 3 * d; qa.slab = w.pl_slab; qa.counters = w.pl_cnt;
    if (gemv_pl_forward(L.attn_p, qa, st)) return 1;
    DecodeAttnArgs da;
    da.qkv_part = w.qkvd; da.parts = 1; da.part_rows = B; da.qkv_bias = nullptr;
    da.kcache = w.kcache + li * per_layer; da.vcache = w.vcache + li * per_layer; da.kv16 = kv_fmt; da.out_row = w.attrow; da.kstart = w.kstart;
    da.st = w.state; da.B = B; da.H = cfg.heads; da.Smax = w.Smax; da.d = d; da.scale = 0.125f;
    da.nsplit = decode_attn_nsplit(B, cfg.heads); da.part = w.attn_part; da.cnt = w.attn_cnt;
    da.pos_hint = tl_prof_pos;
    if (decode_attn_forward(da, st)) return 1;
    GemvPLArgs pa;      // x += c_proj(attn) + b (in place: a lane reads and writes only its own elements of x), + the row statistics of the new x
    pa.x = w.attrow; pa.ldx = d; pa.rows = B; pa.bias = L.proj_l.bias; pa.res = w.xrow; pa.y = w.xrow; pa.ldy = d; pa.stats_out = w.stats;
    pa.slab = w.pl_slab; pa.counters = w.pl_cnt;
    if (gemv_pl_forward(L.proj_p, pa, st)) return 1;
    GemvPLArgs fa;      // ff = gelu_new(c_fc(LN2(x)) + b)
    fa.x = w.xrow; fa.ldx = d; fa.rows = B; fa.colsum = L.fc_u; fa.bias = L.fc_c; fa.stats_in = w.stats; fa.stats_tiles = T; fa.act = 1; fa.y = w.ffrow; fa.ldy = 4 * d;
    fa.slab = w.pl_slab; fa.counters = w.pl_cnt;
    if (gemv_pl_forward(L.fc_p, fa, st)) return 1;
    GemvPLArgs fb;      // x += mlp.c_proj(ff) + b
    fb.x = w.ffrow; fb.ldx = 4 * d; fb.rows = B; fb.bias = L.fc2_l.bias; fb.res = w.xrow; fb.y = w.xrow; fb.ldy = d; fb.stats_out = w.stats;
    fb.slab = w.pl_slab; fb.counters = w.pl_cnt;
    if (gemv_pl_forward(L.fc2_p, fb, st)) return 1;
  }
  if (head_and_sample(w, B, w.xrow, d, false, penalty, codes, codes_ld, logits_base, st)) return 1;
  return fused ? 0 : advance_state(w.state, st);
}

int GPTModel::decode_step(const Buffers& w, int B, float penalty, long long* codes, int codes_ld, float* logits_base,
                          hipStream_t st) {
  if (use_pl(B)) return decode_step_pl(w, B, penalty, codes, codes_ld, logits_base, st);
  const int d = cfg.model_dim;
  const size_t per_layer = kv_layer_bytes(B, w.Smax);
  const bool fused = fused_tail(B);      // greedy: the previous step's sampler has written this step's x and advanced the step scalars
  if (!fused && embed_step(w.xd, B, d, mel_emb, mel_pos, w.cur_tok, w.state, st)) return 1;
  for (int li = 0; li < cfg.layers; ++li) {
    const GPTLayer& L = layers[li];
    GemvFXArgs qa;      // qkv = c_attn(LN1(x)) + b, row-major for the attention kernel
    qa.xf = w.xd; qa.rows = B; qa.colsum = L.attn_u; qa.bias = L.attn_c; qa.y = w.qkvd; qa.ldy = 3 * d;
    if (gemv_fx_forward(L.attn_g, qa, st)) return 1;
    DecodeAttnArgs da;
    da.qkv_part = w.qkvd; da.parts = 1; da.part_rows = B; da.qkv_bias = nullptr;
    da.kcache = w.kcache + li * per_layer; da.vcache = w.vcache + li * per_layer; da.kv16 = kv_fmt; da.out = w.attd; da.kstart = w.kstart;
    da.st = w.state; da.B = B; da.H = cfg.heads; da.Smax = w.Smax; da.d = d; da.scale = 0.125f;
    da.nsplit = decode_attn_nsplit(B, cfg.heads); da.part = w.attn_part; da.cnt = w.attn_cnt;
    da.pos_hint = tl_prof_pos;
    if (decode_attn_forward(da, st)) return 1;
    GemvFXArgs pa;      // x += c_proj(attn) + b  (in place: a thread reads and writes only its own element of x)
    pa.xf = w.attd; pa.rows = B; pa.bias = L.proj_l.bias; pa.res = w.xd; pa.y = w.xd; pa.y_frag = 1;
    if (gemv_fx_forward(L.proj_g, pa, st)) return 1;
    GemvFXArgs fa;      // ff = gelu_new(c_fc(LN2(x)) + b)
    fa.xf = w.xd; fa.rows = B; fa.colsum = L.fc_u; fa.bias = L.fc_c; fa.act = 1; fa.y = w.ffd; fa.y_frag = 1;
    if (gemv_fx_forward(L.fc_g, fa, st)) return 1;
    GemvFXArgs fb;      // x += mlp.c_proj(ff) + b
    fb.xf = w.ffd; fb.rows = B;
    const int ksb = gemv_fx_ksb(L.fc2_g.N, L.fc2_g.K);
    if (ksb > 1) {          // K split across workgroups (all 256 CUs stream); the partial sums are combined in a fixed order by the
                            // last workgroup of each column tile to arrive (no combine launch)
      fb.ksb = ksb; fb.slab = w.slab; fb.ksb_counters = w.ksb_cnt;
      fb.bias = L.fc2_l.bias; fb.res = w.xd; fb.y = w.xd; fb.y_frag = 1;
      if (gemv_fx_forward(L.fc2_g, fb, st)) return 1;
    } else {
      fb.bias = L.fc2_l.bias; fb.res = w.xd; fb.y = w.xd; fb.y_frag = 1;
      if (gemv_fx_forward(L.fc2_g, fb, st)) return 1;
    }
  }
  if (head_and_sample(w, B, w.xd, d, true, penalty, codes, codes_ld, logits_base, st)) return 1;
  return fused ? 0 : advance_state(w.state, st);
}

int GPTModel::generate(const float* inputs_embeds, const int* pad_left_host, int B, int P, int max_new, float penalty,
                       const idxtts_sampling* sampling, long long* codes, int* n_steps_out, float* logits_out, void* ws, size_t ws_bytes, int use_graph,
                       hipStream_t user_stream, const long long* forced) {
  IDX_CHECK(inputs_embeds && codes && n_steps_out, "null pointer");
  GenScope gen_scope(this);
  struct ForcedScope {      // the forced tokens apply to this call only, whatever path it returns on
    ForcedScope(const long long* f, int ld) { tl_forced = f; tl_forced_ld = ld; }
    ~ForcedScope() { tl_forced = nullptr; tl_forced_ld = 0; }
  } forced_scope(forced, max_new);
  IDX_CHECK(!forced || !(sampling && sampling->mode != 0), "teacher forcing is a greedy-mode instrument");
  // The legacy default stream cannot be captured into a graph: run on a private stream, ordered after
  // everything already queued by the caller (the call ends with a host sync anyway: n_steps is a host value).
  hipStream_t st = user_stream;
  if (user_stream == nullptr) {
    if (!own_stream) IDX_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    IDX_HIP(hipStreamSynchronize(user_stream));
    st = own_stream;
  }
  IDX_CHECK(B > 0 && B <= 64 && P > 0 && max_new > 0, "shape (1 <= B <= 64)");
  samp = idxtts_sampling{0, 1.0f, 0, 1.0f, nullptr, 0};
  if (sampling && sampling->mode != 0) {
    IDX_CHECK(sampling->mode == SAMPLE_HF || sampling->mode == SAMPLE_ACCEL, "sampling mode");
    IDX_CHECK(sampling->temperature > 0.0f, "sampling needs a positive temperature");
    IDX_CHECK(sampling->top_p >= 1.0f || (sampling->top_k > 0 && sampling->top_k <= 1024), "top-p needs 0 < top_k <= 1024");
    samp = *sampling;
  }
  const int d = cfg.model_dim, V = cfg.number_mel_codes, S = P + 1;
  IDX_CHECK(max_new + 1 < cfg.mel_pos_len, "max_new_tokens exceeds the mel position table");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, S, max_new), "workspace too small");
  Buffers w = carve(ws, B, S, max_new);
  long long* const user_codes = codes;
  codes = w.codes;           // every launch writes here (a stable address: the captured decode step can be kept); copied out at the end
  IDX_HIP(hipMemcpyAsync(codes, user_codes, (size_t)B * max_new * sizeof(long long), hipMemcpyDeviceToDevice, st));      // the caller's pad fill

  // ---- per-call state ----
  std::vector<int> kstart(B, 0);
  if (pad_left_host) for (int b = 0; b < B; ++b) kstart[b] = pad_left_host[b];
  for (int b = 0; b < B; ++b) IDX_CHECK(kstart[b] >= 0 && kstart[b] < P, "pad_left out of range");
  IDX_HIP(hipMemcpyAsync(w.kstart, kstart.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemsetAsync(w.finished, 0, B * sizeof(int), st));
  IDX_HIP(hipMemsetAsync(w.ksb_cnt, 0, (size_t)cdiv(d, 16) * sizeof(unsigned), st));
  IDX_HIP(hipMemsetAsync(w.attn_cnt, 0, (size_t)B * cfg.heads * sizeof(unsigned), st));
  IDX_HIP(hipMemsetAsync(static_cast<char*>(ws) + w.frag_off, 0, w.frag_bytes, st));   // padding rows of the fragment images
  // input_ids of the reference = fake prefix of 1s + start_mel_token: both count for the repetition penalty
  std::vector<unsigned char> seen((size_t)B * V, 0);
  for (int b = 0; b < B; ++b) { seen[(size_t)b * V + 1] = 1; seen[(size_t)b * V + cfg.start_mel_token] = 1; }
  IDX_HIP(hipMemcpyAsync(w.seen, seen.data(), seen.size(), hipMemcpyHostToDevice, st));
  DecodeState s0{P, 1, 0, 0};
  IDX_HIP(hipMemcpyAsync(w.state, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));   // host staging buffers go out of scope below

  // ---- prefill: x = [inputs_embeds | mel_emb[start] + mel_pos[0]] ----
  IDX_HIP(hipMemcpy2DAsync(w.x, (size_t)S * d * sizeof(float), inputs_embeds, (size_t)P * d * sizeof(float),
                           (size_t)P * d * sizeof(float), B, hipMemcpyDeviceToDevice, st));
  {
    std::vector<int> tok(B, cfg.start_mel_token);
    IDX_HIP(hipMemcpyAsync(w.cur_tok, tok.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    IDX_HIP(hipStreamSynchronize(st));
    DecodeState zero{0, 0, 0, 0};
    (void)zero;
    // reuse embed_step with a temporary state whose mel_pos = 0: state currently has mel_pos = 1, so gather directly
    GatherArgs ga;
    ga.out = w.x + (size_t)P * d; ga.ld_out = S * d; ga.d = d;
    ga.table[0] = mel_emb; ga.idx[0] = w.cur_tok;           // start_mel_token rows
    ga.table[1] = mel_pos; ga.idx[1] = w.finished;          // all zeros -> mel position 0
    if (gather_sum_rows(ga, B, st)) return 1;
  }
  for (int li = 0; li < cfg.layers; ++li)
    if (layer_full(li, w, B, S, w.kstart, true, st)) return 1;
  {
    // last position of every row
    if (head_and_sample(w, B, w.x + (size_t)(S - 1) * d, S * d, false, penalty, codes, max_new, logits_out, st)) return 1;
    if (!fused_tail(B) && advance_state(w.state, st)) return 1;      // (the fused sampler of the plane-GEMV path has advanced already)
  }

  // ---- decode ----
  struct GraphGuard {      // a graph that is not kept: released on every return path (error paths after the capture included)
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    ~GraphGuard() { if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
  } gg;
  struct SlotLease {       // a cached graph in use by this call
    GPTModel* m = nullptr; int idx = -1;
    ~SlotLease() { if (m && idx >= 0) { std::lock_guard<std::mutex> l(m->graph_mu); m->graph_cache[idx].in_use = false; } }
  } lease;
  hipGraphExec_t exec = nullptr;
  const bool graph_ok = use_graph && !logits_out && !forced && !prof_enabled();
  const bool cacheable = samp.mode == 0 && !tl_beam;      // greedy: nothing call-specific is baked into the launches
  int n_first = 1;
  if (graph_ok && max_new > 2) {
    // step 1 runs eagerly (first-use function attributes are set outside the capture), steps >= 2 replay
    if (decode_step(w, B, penalty, codes, max_new, nullptr, st)) return 1;
    n_first = 2;
    if (cacheable) {
      std::lock_guard<std::mutex> l(graph_mu);
      for (size_t i = 0; i < graph_cache.size(); ++i) {
        GraphSlot& g = graph_cache[i];
        if (!g.in_use && g.ws == ws && g.ws_bytes == ws_bytes && g.B == B && g.S == S && g.max_new == max_new && g.penalty == penalty && g.kv16 == kv_fmt && g.geom == (get_decode_geometry() | (get_decode_plane_rows() << 1))) {
          g.in_use = true; g.stamp = ++graph_stamp; exec = g.exec; lease.m = this; lease.idx = (int)i;
          break;
        }
      }
    }
    if (!exec) {
      IDX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = decode_step(w, B, penalty, codes, max_new, nullptr, st);
      hipError_t e = hipStreamEndCapture(st, &gg.graph);
      if (rc) return 1;
      IDX_HIP(e);
      IDX_HIP(hipGraphInstantiate(&gg.exec, gg.graph, nullptr, nullptr, 0));
      exec = gg.exec;
      if (cacheable) {      // keep it: hand the objects over to a cache slot (a free one, or the least recently used idle one)
        std::lock_guard<std::mutex> l(graph_mu);
        int slot = -1;
        // an idle entry on the same workspace address describes launches that can no longer be replayed safely: replace it
        for (size_t i = 0; i < graph_cache.size() && slot < 0; ++i) if (!graph_cache[i].in_use && graph_cache[i].ws == ws) slot = (int)i;
        if (slot < 0 && graph_cache.size() < GRAPH_CACHE_MAX) { graph_cache.emplace_back(); slot = (int)graph_cache.size() - 1; }
        if (slot < 0) {
          for (size_t i = 0; i < graph_cache.size(); ++i)
            if (!graph_cache[i].in_use && (slot < 0 || graph_cache[i].stamp < graph_cache[slot].stamp)) slot = (int)i;
        }
        if (slot >= 0) {
          GraphSlot& g = graph_cache[slot];
          if (g.exec) (void)hipGraphExecDestroy(g.exec);
          if (g.graph) (void)hipGraphDestroy(g.graph);
          g.ws = ws; g.ws_bytes = ws_bytes; g.B = B; g.S = S; g.max_new = max_new; g.penalty = penalty; g.kv16 = kv_fmt; g.geom = get_decode_geometry() | (get_decode_plane_rows() << 1);
          g.graph = gg.graph; g.exec = gg.exec; g.stamp = ++graph_stamp; g.in_use = true;
          gg.graph = nullptr; gg.exec = nullptr;
          lease.m = this; lease.idx = slot;
        }
      }
    }
  }
  std::vector<int> fin(B);
  int steps_done = n_first;
  for (int n = n_first; n < max_new; ++n) {
    if (exec) {
      IDX_HIP(hipGraphLaunch(exec, st));
    } else {
      float* lo = logits_out ? logits_out + (size_t)n * B * V : nullptr;
      tl_prof_pos = S + n;      // keys this step reads (profiler accounting of the decode attention)
      if (decode_step(w, B, penalty, codes, max_new, lo, st)) return 1;
      tl_prof_pos = 0;
    }
    steps_done = n + 1;
    if ((n & 15) == 15 || n + 1 == max_new) {     // all rows finished? (HF stops there; later columns would be pad)
      IDX_HIP(hipMemcpyAsync(fin.data(), w.finished, B * sizeof(int), hipMemcpyDeviceToHost, st));
      IDX_HIP(hipStreamSynchronize(st));
      bool all = true;
      for (int b = 0; b < B; ++b) all = all && fin[b];
      if (all) break;
    }
  }
  // exact HF length: generation stops at the first step after which every row has emitted the stop token
  std::vector<long long> hc((size_t)B * max_new);
  IDX_HIP(hipMemcpyAsync(user_codes, codes, hc.size() * sizeof(long long), hipMemcpyDeviceToDevice, st));
  IDX_HIP(hipMemcpyAsync(hc.data(), codes, hc.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipStreamSynchronize(st));
  int n_steps = steps_done;
  int worst = 0;
  bool every_row_stops = true;
  for (int b = 0; b < B; ++b) {
    int first = -1;
    for (int s = 0; s < steps_done; ++s)
      if (hc[(size_t)b * max_new + s] == cfg.stop_mel_token) { first = s; break; }
    if (first < 0) every_row_stops = false;
    else worst = std::max(worst, first + 1);
  }
  if (every_row_stops && !forced) n_steps = worst;      // forced: every step that ran is reported (codes = the rows' own choices)
  *n_steps_out = n_steps;
  return 0;
}

int GPTModel::latent(const float* emb, const int* pad_left_host, int B, int S, int mel_start, int M, float* latent_out, void* ws,
                     size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(emb && latent_out, "null pointer");
  IDX_CHECK(B > 0 && S > 0 && M > 0 && mel_start >= 0 && mel_start + M <= S, "shape");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, S, 0), "workspace too small");
  const int d = cfg.model_dim;
  Buffers w = carve(ws, B, S, 0);
  const int* kstart = nullptr;
  if (pad_left_host) {      // rows with shorter texts are left-padded; padded keys are masked exactly like the decode prompt
    for (int b = 0; b < B; ++b) IDX_CHECK(pad_left_host[b] >= 0 && pad_left_host[b] < mel_start, "pad_left out of range");
    IDX_HIP(hipMemcpyAsync(w.kstart, pad_left_host, B * sizeof(int), hipMemcpyHostToDevice, st));
    IDX_HIP(hipStreamSynchronize(st));
    kstart = w.kstart;
  }
  IDX_HIP(hipMemcpyAsync(w.x, emb, (size_t)B * S * d * sizeof(float), hipMemcpyDeviceToDevice, st));
  for (int li = 0; li < cfg.layers; ++li)
    if (layer_full(li, w, B, S, kstart, false, st)) return 1;
  RowsNormArgs n;     // final_norm(ln_f(h)) on the mel rows only (model_v2.py:611, 723)
  n.x_in = w.x + (size_t)mel_start * d; n.ld_in = d; n.in_rows_per_batch = M; n.in_batch_stride = (long)S * d;
  n.y = latent_out; n.ld_y = d; n.M = B * M; n.d = d; n.mode = NORM_LN_LN;
  n.g1 = lnf_g; n.b1 = lnf_b; n.g2 = fn_g; n.b2 = fn_b;
  return rows_norm_forward(n, st);
}

int GPTModel::embed(float* out, int rows, const int* text_ids, const int* text_pos_idx, const int* mel_ids, const int* mel_pos_idx,
                    const float* extra, const int* extra_idx, hipStream_t st) {
  GatherArgs ga;
  ga.out = out; ga.ld_out = cfg.model_dim; ga.d = cfg.model_dim;
  ga.table[0] = text_emb; ga.idx[0] = text_ids;
  ga.table[1] = text_pos; ga.idx[1] = text_pos_idx;
  ga.table[2] = mel_emb; ga.idx[2] = mel_ids;
  ga.table[3] = mel_pos; ga.idx[3] = mel_pos_idx;
  ga.table[4] = extra; ga.idx[4] = extra_idx;
  ga.table_rows[0] = cfg.number_text_tokens + 1; ga.table_rows[1] = cfg.text_pos_len;
  ga.table_rows[2] = cfg.number_mel_codes; ga.table_rows[3] = cfg.mel_pos_len;
  // one flag slot per call in flight: embed() runs concurrently on several streams (serving.BatchPipeline's lanes), and a shared
  // flag could be cleared by one call's memset before another call has read it
  int* const flag = oob_flag + (oob_next.fetch_add(1, std::memory_order_relaxed) % OOB_SLOTS);
  ga.oob = flag;
  IDX_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
  if (gather_sum_rows(ga, rows, st)) return 1;
  int bad = 0;       // the reference's nn.Embedding raises IndexError on such an id (model_v2.py:759-760); fail as loudly
  IDX_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipStreamSynchronize(st));
  static const char* names[] = {"", "text token id", "text position", "mel code", "mel position", "conditioning row"};
  if (bad) IDX_FAIL(std::string("embedding index out of range: ") + names[bad < 6 ? bad : 0]);
  return 0;
}

}  // namespace idxtts
