// extern "C" surface of libidxtts_hip (declared in include/idxtts.h).  Nothing here throws.
#include <cstring>
#include <new>

#include "../../include/idxtts.h"
#include "bigvgan.h"
#include "cond.h"
#include "semantic.h"
#include "codec.h"
#include "audio.h"
#include "campplus.h"
#include "conv1d.h"
#include "ctx.h"
#include "gpt.h"
#include "s2mel.h"

namespace idxtts {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(const char* file, int line, const std::string& msg) {
  const char* base = std::strrchr(file, '/');
  g_last_error = std::string(base ? base + 1 : file) + ":" + std::to_string(line) + ": " + msg;
  return 1;
}

int aa_act_forward(void* y, const void* x, const float* up_f, const float* down_f, const float* log_alpha,
                   const float* log_beta, int B, int C, int T, hipStream_t stream, const int* lens = nullptr, int len_mul = 1, int dtype = 0);

}  // namespace idxtts

using namespace idxtts;

struct idxtts_conv1d {
  ConvWeights w;
  DeviceArena arena;
  int Cout = 0;
};

struct idxtts_linear {
  LinearWeights w;
  DeviceArena arena;
};

#define API_BEGIN try {
#define API_END                                                        \
  }                                                                    \
  catch (const std::exception& e) { return fail(__FILE__, __LINE__, std::string("exception: ") + e.what()); } \
  catch (...) { return fail(__FILE__, __LINE__, "unknown exception"); }

extern "C" {

int idxtts_version(void) { return 100; }

const char* idxtts_last_error(void) { return g_last_error.c_str(); }

int idxtts_aa_act_fwd(void* out, const void* in, const float* up_filter, const float* down_filter,
                      const float* log_alpha, const float* log_beta, int B, int C, int T, int dtype, void* stream) {
  API_BEGIN
  IDX_CHECK(dtype == IDXTTS_DTYPE_F32 || dtype == IDXTTS_DTYPE_F16 || dtype == IDXTTS_DTYPE_BF16, "dtype must be IDXTTS_DTYPE_F32 / F16 / BF16");
  IDX_CHECK(B >= 0 && C >= 0 && T >= 0, "negative shape");
  return aa_act_forward(out, in, up_filter, down_filter, log_alpha, log_beta, B, C, T, static_cast<hipStream_t>(stream), nullptr, 1, dtype);
  API_END
}

// copies `n` floats from a host-or-device pointer into a host vector
static int fetch(const float* src, size_t n, std::vector<float>* dst) {
  dst->resize(n);
  if (!n) return 0;
  const hipError_t e = hipMemcpy(dst->data(), src, n * sizeof(float), hipMemcpyDefault);      // host or device source
  if (e == hipErrorNoDevice) {       // no GPU in this process (host-side staging / tests): every pointer is a host pointer
    (void)hipGetLastError();
    memcpy(dst->data(), src, n * sizeof(float));
    return 0;
  }
  IDX_HIP(e);
  return 0;
}

int idxtts_conv1d_create(const float* weight, const float* bias, int Cout, int Cin, int K, int transposed_stride,
                         idxtts_conv1d** out) {
  API_BEGIN
  IDX_CHECK(weight && out, "null pointer");
  IDX_CHECK(Cout > 0 && Cin > 0 && K > 0 && transposed_stride >= 1, "bad shape");
  std::unique_ptr<idxtts_conv1d> c(new idxtts_conv1d());
  std::vector<float> hw, hb;
  if (fetch(weight, (size_t)Cout * Cin * K, &hw)) return 1;
  c->Cout = Cout;
  c->w.Cin = Cin;
  c->w.nchunk = cdiv(Cin, CONV_KC);
  c->w.ups = transposed_stride;
  if (transposed_stride > 1) {
    IDX_CHECK(K == 2 * transposed_stride, "ConvTranspose1d kernel must be 2*stride");
    c->w.M = Cout * transposed_stride;
    c->w.K = 3;
  } else {
    c->w.M = Cout;
    c->w.K = K;
  }
  std::vector<float> packed(conv_packed_floats(c->w.M, Cin, c->w.K));
  if (transposed_stride > 1) pack_conv_transpose1d(packed.data(), hw.data(), Cin, Cout, K, transposed_stride);
  else pack_conv1d(packed.data(), hw.data(), Cout, Cin, K);
  float* d = nullptr;
  if (c->arena.upload(packed.data(), packed.size(), &d)) return 1;
  c->w.wp = d;
  {
    std::vector<float> p16(packed.size());      // hi + lo bf16 = 4 bytes per weight, as the fp32 pack
    pack_conv_bf16x3(p16.data(), packed.data(), packed.size() / CONV_SUB);
    if (c->arena.upload(p16.data(), p16.size(), &d)) return 1;
    c->w.wp16 = d;
  }
  if (bias) {
    if (fetch(bias, Cout, &hb)) return 1;
    if (c->arena.upload(hb.data(), hb.size(), &d)) return 1;
    c->w.bias = d;
  }
  *out = c.release();
  return 0;
  API_END
}

int idxtts_conv1d_fwd(const idxtts_conv1d* conv, const float* x, float* y, const float* residual, int B, int T,
                      int dilation, int pad_left, int pad_mode, float scale, int accumulate, void* stream) {
  API_BEGIN
  IDX_CHECK(conv, "null handle");
  if (B == 0 || T == 0) return 0;
  ConvArgs a;
  a.x = x; a.y = y; a.res = residual; a.B = B; a.T = T; a.dil = dilation; a.pad_left = pad_left;
  a.pad_mode = pad_mode; a.scale = scale; a.accum = accumulate;
  return conv1d_forward(conv->w, a, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_conv1d_destroy(idxtts_conv1d* conv) {
  delete conv;
  return 0;
}

int idxtts_ctx_load_tensor(idxtts_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim) {
  API_BEGIN
  IDX_CHECK(ctx && name && shape, "null pointer");
  IDX_CHECK(!ctx->finalized, "context already finalized");
  IDX_CHECK(ndim >= 0 && ndim <= 8, "ndim");
  const std::string key(name);
  if (!ctx->model->accepts(key)) IDX_FAIL("unknown tensor name '" + key + "'");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const int64_t n = t.numel();
  IDX_CHECK(n >= 0, "negative numel");
  IDX_CHECK(n == 0 || data, "null data");
  if (fetch(data, (size_t)n, &t.data)) return 1;
  ctx->tensors[key] = std::move(t);
  return 0;
  API_END
}

int idxtts_ctx_finalize(idxtts_ctx* ctx) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(!ctx->finalized, "context already finalized");
  if (ctx->model->finalize(ctx->tensors, ctx->arena)) return 1;
  ctx->tensors.clear();
  ctx->finalized = true;
  return 0;
  API_END
}

int idxtts_ctx_get_tensor(idxtts_ctx* ctx, const char* name, float* host_out, size_t capacity) {
  API_BEGIN
  IDX_CHECK(ctx && name && host_out, "null pointer");
  IDX_CHECK(!ctx->finalized, "staged tensors are released by finalize");
  auto it = ctx->tensors.find(name);
  if (it == ctx->tensors.end()) IDX_FAIL(std::string("no staged tensor '") + name + "'");
  IDX_CHECK(capacity >= it->second.data.size(), "output buffer too small");
  std::copy(it->second.data.begin(), it->second.data.end(), host_out);
  return 0;
  API_END
}

int idxtts_ctx_destroy(idxtts_ctx* ctx) {
  delete ctx;
  return 0;
}

float idxtts_fp8_e4m3_decode(unsigned char code) { return fp8_e4m3_decode(code); }
unsigned char idxtts_fp8_e4m3_encode(float v) { return fp8_e4m3_encode(v); }

int idxtts_bigvgan_create(const idxtts_bigvgan_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new BigVGANModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

size_t idxtts_bigvgan_workspace_bytes(const idxtts_ctx* ctx, int B, int Tm) {
  if (!ctx || B <= 0 || Tm <= 0) return 0;
  auto* m = dynamic_cast<const BigVGANModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, Tm) : 0;
}

int idxtts_bigvgan_fwd(idxtts_ctx* ctx, const float* mel, float* wav, int B, int Tm, void* workspace,
                       size_t workspace_bytes, int clamp, int stage_idx, float* stage_out, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<BigVGANModel*>(ctx->model.get());
  IDX_CHECK(m, "not a BigVGAN context");
  IDX_CHECK(B >= 0 && Tm >= 0, "negative shape");
  return m->forward(mel, wav, B, Tm, workspace, workspace_bytes, clamp, stage_idx, stage_out, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_bigvgan_fwd_ragged(idxtts_ctx* ctx, const float* mel, const int* mel_lengths, float* wav, int B, int Tm, void* workspace,
                              size_t workspace_bytes, int clamp, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx && ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<BigVGANModel*>(ctx->model.get());
  IDX_CHECK(m, "not a BigVGAN context");
  IDX_CHECK(B >= 0 && Tm >= 0 && mel_lengths, "shape / lengths");
  return m->forward(mel, wav, B, Tm, workspace, workspace_bytes, clamp, 0, nullptr, static_cast<hipStream_t>(stream), mel_lengths);
  API_END
}


int idxtts_linear_create(const float* weight, const float* bias, int N, int K, int weight_is_kn, idxtts_linear** out) {
  API_BEGIN
  IDX_CHECK(weight && out && N > 0 && K > 0, "bad arguments");
  std::unique_ptr<idxtts_linear> l(new idxtts_linear());
  std::vector<float> hw, hb;
  if (fetch(weight, (size_t)N * K, &hw)) return 1;
  std::vector<float> packed(linear_packed_floats(N, K));
  if (weight_is_kn) pack_linear_kn(packed.data(), hw.data(), K, N);
  else pack_linear(packed.data(), hw.data(), N, K);
  float* d = nullptr;
  if (l->arena.upload(packed.data(), packed.size(), &d)) return 1;
  l->w.wp = d; l->w.N = N; l->w.K = K;
  {
    std::vector<float> wnk(hw);
    if (weight_is_kn)
      for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) wnk[(size_t)n * K + k] = hw[(size_t)k * N + n];
    std::vector<float> p16((linear_bf16x3_packed_bytes(N, K) + 3) / 4);
    pack_linear_bf16x3(p16.data(), wnk.data(), N, K);
    float* d16 = nullptr;
    if (l->arena.upload(p16.data(), p16.size(), &d16)) return 1;
    l->w.wp16 = d16;
  }
  if (bias) {
    if (fetch(bias, N, &hb)) return 1;
    if (l->arena.upload(hb.data(), hb.size(), &d)) return 1;
    l->w.bias = d;
  }
  *out = l.release();
  return 0;
  API_END
}

int idxtts_linear_fwd(const idxtts_linear* lin, const float* x, int ldx, float* y, int ldy, const float* residual, int ldr, int M,
                      int act, int bf16x3, void* stream) {
  API_BEGIN
  IDX_CHECK(lin, "null handle");
  GemmArgs a;
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.res = residual; a.ldr = ldr; a.M = M; a.act = act;
  return bf16x3 ? gemm_bf16x3_forward(lin->w, a, static_cast<hipStream_t>(stream)) : gemm_tn_forward(lin->w, a, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_linear_destroy(idxtts_linear* lin) {
  delete lin;
  return 0;
}

int idxtts_attention_fwd(const float* q, const float* k, const float* v, float* o, long q_batch_stride, int q_token_stride,
                         long kv_batch_stride, int kv_token_stride, long o_batch_stride, int o_token_stride, int B, int H,
                         int Sq, int Sk, int causal, const int* kstart, const int* kend, float scale, void* stream) {
  API_BEGIN
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.q_bs = q_batch_stride; a.k_bs = a.v_bs = kv_batch_stride; a.o_bs = o_batch_stride;
  a.q_ts = q_token_stride; a.k_ts = a.v_ts = kv_token_stride; a.o_ts = o_token_stride;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.kstart = kstart; a.kend = kend; a.scale = scale;
  return flash_attn_forward(a, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_attention_relkey_fwd(const float* q, const float* k, const float* v, float* o, long batch_stride, int token_stride,
                                long o_batch_stride, int o_token_stride, int B, int H, int S, const int* kend, float scale,
                                const float* rel_key, int rel_left, int rel_right, int split_bf16, void* stream) {
  API_BEGIN
  IDX_CHECK(rel_key, "rel_key is null");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.q_bs = a.k_bs = a.v_bs = batch_stride; a.o_bs = o_batch_stride;
  a.q_ts = a.k_ts = a.v_ts = token_stride; a.o_ts = o_token_stride;
  a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.causal = 0; a.kend = kend; a.scale = scale;
  a.rel_key = rel_key; a.rel_left = rel_left; a.rel_right = rel_right; a.split_bf16 = split_bf16 ? 1 : 0;
  return flash_attn_forward(a, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_attention_bf16x3_fwd(const float* q, const float* k, const float* v, float* o, long q_batch_stride, int q_token_stride,
                         long kv_batch_stride, int kv_token_stride, long o_batch_stride, int o_token_stride, int B, int H,
                         int Sq, int Sk, int causal, const int* kstart, const int* kend, float scale, void* stream) {
  API_BEGIN
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.q_bs = q_batch_stride; a.k_bs = a.v_bs = kv_batch_stride; a.o_bs = o_batch_stride;
  a.q_ts = q_token_stride; a.k_ts = a.v_ts = kv_token_stride; a.o_ts = o_token_stride;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.kstart = kstart; a.kend = kend; a.scale = scale; a.split_bf16 = 1;
  return flash_attn_forward(a, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int M, int d, float eps, void* stream) {
  API_BEGIN
  RowsNormArgs n;
  n.x_in = x; n.ld_in = d; n.y = y; n.ld_y = d; n.M = M; n.d = d; n.mode = NORM_LN; n.eps = eps; n.g1 = gamma; n.b1 = beta;
  return rows_norm_forward(n, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_gpt_create(const idxtts_gpt_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new GPTModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

int idxtts_gpt_quantize_weights(idxtts_ctx* ctx, int format) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(!ctx->finalized, "quantise before idxtts_ctx_finalize");
  auto* m = dynamic_cast<GPTModel*>(ctx->model.get());
  IDX_CHECK(m, "not a GPT context");
  return m->quantize_weights(ctx->tensors, format);
  API_END
}

int idxtts_gpt_set_kv_format(idxtts_ctx* ctx, int format) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(format == 0 || format == 1, "KV cache format: 0 (fp32) or 1 (bf16)");
  auto* m = dynamic_cast<GPTModel*>(ctx->model.get());
  IDX_CHECK(m, "not a GPT context");
  // a generation in flight has carved its workspace for the current format: switching under it would overrun the cache
  IDX_CHECK(m->kv_fmt == format || m->generating.load() == 0, "a generation is in flight on this context: switch the KV format between generations");
  m->kv_fmt = format;      // cached decode graphs carry the format in their key; workspace sizes follow idxtts_gpt_workspace_bytes
  return 0;
  API_END
}

int idxtts_gpt_get_kv_format(const idxtts_ctx* ctx) {
  if (!ctx) return -1;
  auto* m = dynamic_cast<const GPTModel*>(ctx->model.get());
  return m ? m->kv_fmt : -1;
}

int idxtts_gpt_graph_cache_entries(idxtts_ctx* ctx) {
  if (!ctx || !ctx->finalized) return -1;
  auto* m = dynamic_cast<GPTModel*>(ctx->model.get());
  if (!m) return -1;
  std::lock_guard<std::mutex> l(m->graph_mu);
  return (int)m->graph_cache.size();
}

size_t idxtts_gpt_workspace_bytes(const idxtts_ctx* ctx, int B, int S, int max_new_tokens) {
  if (!ctx || B <= 0 || S <= 0 || max_new_tokens < 0 || !ctx->finalized) return 0;
  auto* m = dynamic_cast<const GPTModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, S, max_new_tokens) : 0;
}

#define GPT_MODEL(ctx)                                        \
  IDX_CHECK(ctx, "null ctx");                                 \
  IDX_CHECK(ctx->finalized, "context not finalized");         \
  auto* m = dynamic_cast<GPTModel*>(ctx->model.get());        \
  IDX_CHECK(m, "not a GPT context")

int idxtts_gpt_embed(idxtts_ctx* ctx, float* out, int rows, const int* text_ids, const int* text_pos_idx, const int* mel_ids,
                     const int* mel_pos_idx, const float* extra, const int* extra_idx, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  return m->embed(out, rows, text_ids, text_pos_idx, mel_ids, mel_pos_idx, extra, extra_idx, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_gpt_generate(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                        float repetition_penalty, long long* codes, int* n_steps, float* logits_out, void* workspace,
                        size_t workspace_bytes, int use_graph, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  return m->generate(inputs_embeds, pad_left, B, P, max_new_tokens, repetition_penalty, nullptr, codes, n_steps, logits_out, workspace,
                     workspace_bytes, use_graph, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_gpt_generate_forced(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                               float repetition_penalty, const long long* forced_codes, long long* codes, int* n_steps, float* logits_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  IDX_CHECK(forced_codes, "null forced_codes");
  return m->generate(inputs_embeds, pad_left, B, P, max_new_tokens, repetition_penalty, nullptr, codes, n_steps, logits_out, workspace,
                     workspace_bytes, 0, static_cast<hipStream_t>(stream), forced_codes);
  API_END
}

int idxtts_gpt_generate_sampled(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                                float repetition_penalty, const idxtts_sampling* sampling, long long* codes, int* n_steps,
                                float* logits_out, void* workspace, size_t workspace_bytes, int use_graph, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  IDX_CHECK(sampling, "null sampling configuration");
  return m->generate(inputs_embeds, pad_left, B, P, max_new_tokens, repetition_penalty, sampling, codes, n_steps, logits_out, workspace,
                     workspace_bytes, use_graph, static_cast<hipStream_t>(stream));
  API_END
}

size_t idxtts_gpt_beam_workspace_bytes(const idxtts_ctx* ctx, int B, int num_beams, int S, int max_new_tokens) {
  if (!ctx || !ctx->finalized || B <= 0 || num_beams < 2 || S <= 0 || max_new_tokens <= 0) return 0;
  auto* m = dynamic_cast<const GPTModel*>(ctx->model.get());
  return m ? m->beam_workspace_bytes(B, num_beams, S, max_new_tokens) : 0;
}

int idxtts_gpt_generate_beam(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                             float repetition_penalty, const idxtts_beam* beam, long long* codes, int* n_steps, void* workspace,
                             size_t workspace_bytes, int use_graph, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  return m->generate_beam(inputs_embeds, pad_left, B, P, max_new_tokens, repetition_penalty, beam, codes, n_steps, workspace,
                          workspace_bytes, use_graph, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_gpt_latent(idxtts_ctx* ctx, const float* emb, const int* pad_left, int B, int S, int mel_start, int M, float* latent,
                      void* workspace, size_t workspace_bytes, void* stream) {
  API_BEGIN
  GPT_MODEL(ctx);
  return m->latent(emb, pad_left, B, S, mel_start, M, latent, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}


int idxtts_cond_create(const idxtts_cond_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new CondModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

size_t idxtts_cond_workspace_bytes(const idxtts_ctx* ctx, int B, int T) {
  if (!ctx || !ctx->finalized || B <= 0 || T < 3) return 0;
  auto* m = dynamic_cast<const CondModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, T) : 0;
}

int idxtts_cond_forward(idxtts_ctx* ctx, const float* feats, const int* lengths, int B, int T, float* out, void* workspace,
                        size_t workspace_bytes, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<CondModel*>(ctx->model.get());
  IDX_CHECK(m, "not a conditioning-encoder context");
  return m->forward(feats, lengths, B, T, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_w2vbert_create(const idxtts_w2vbert_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new W2VBertModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

size_t idxtts_w2vbert_workspace_bytes(const idxtts_ctx* ctx, int B, int T) {
  if (!ctx || !ctx->finalized || B <= 0 || T <= 0) return 0;
  auto* m = dynamic_cast<const W2VBertModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, T) : 0;
}

int idxtts_w2vbert_forward(idxtts_ctx* ctx, const float* feats, const int* lengths, int B, int T, float* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<W2VBertModel*>(ctx->model.get());
  IDX_CHECK(m, "not a w2v-bert context");
  return m->forward(feats, lengths, B, T, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_repcodec_create(const idxtts_repcodec_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new RepCodecModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

size_t idxtts_repcodec_workspace_bytes(const idxtts_ctx* ctx, int B, int T) {
  if (!ctx || !ctx->finalized || B <= 0 || T <= 0) return 0;
  auto* m = dynamic_cast<const RepCodecModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, T) : 0;
}

int idxtts_repcodec_quantize(idxtts_ctx* ctx, const float* x, int B, int T, long long* indices, float* quantized, void* workspace,
                             size_t workspace_bytes, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<RepCodecModel*>(ctx->model.get());
  IDX_CHECK(m, "not a semantic-codec context");
  return m->quantize(x, B, T, indices, quantized, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_melspec_create(const idxtts_melspec_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new MelSpecModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

int idxtts_melspec_frames(const idxtts_ctx* ctx, int n_samples) {
  if (!ctx || n_samples <= 0) return 0;
  auto* m = dynamic_cast<const MelSpecModel*>(ctx->model.get());
  return m ? m->frames(n_samples) : 0;
}

size_t idxtts_melspec_workspace_bytes(const idxtts_ctx* ctx, int B, int n_samples) {
  if (!ctx || !ctx->finalized || B <= 0 || n_samples <= 0) return 0;
  auto* m = dynamic_cast<const MelSpecModel*>(ctx->model.get());
  return m ? m->workspace_bytes(B, n_samples) : 0;
}

int idxtts_melspec_forward(idxtts_ctx* ctx, const float* audio, int B, int n_samples, float* mel, void* workspace, size_t workspace_bytes,
                           void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<MelSpecModel*>(ctx->model.get());
  IDX_CHECK(m, "not a mel-spectrogram context");
  return m->forward(audio, B, n_samples, mel, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_campplus_create(const idxtts_campplus_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new CamPPlusModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

size_t idxtts_campplus_workspace_bytes(const idxtts_ctx* ctx, int T) {
  if (!ctx || !ctx->finalized || T < 8) return 0;
  auto* m = dynamic_cast<const CamPPlusModel*>(ctx->model.get());
  return m ? m->workspace_bytes(T) : 0;
}

int idxtts_campplus_forward(idxtts_ctx* ctx, const float* feat, int B, int T, float* style, void* workspace, size_t workspace_bytes, void* stream) {
  API_BEGIN
  IDX_CHECK(ctx, "null ctx");
  IDX_CHECK(ctx->finalized, "context not finalized");
  auto* m = dynamic_cast<CamPPlusModel*>(ctx->model.get());
  IDX_CHECK(m, "not a CAMPPlus context");
  return m->forward(feat, B, T, style, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_emovec_merge(float* out, const float* base, const float* emo, float alpha, size_t n, void* stream) {
  API_BEGIN
  return lerp_rows(out, base, emo, alpha, n, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_s2mel_create(const idxtts_s2mel_config* cfg, idxtts_ctx** out) {
  API_BEGIN
  IDX_CHECK(cfg && out, "null pointer");
  std::unique_ptr<idxtts_ctx> ctx(new idxtts_ctx());
  ctx->model.reset(new S2MelModel(*cfg));
  *out = ctx.release();
  return 0;
  API_END
}

#define S2MEL_MODEL(ctx)                                      \
  IDX_CHECK(ctx, "null ctx");                                 \
  IDX_CHECK(ctx->finalized, "context not finalized");         \
  auto* m = dynamic_cast<S2MelModel*>(ctx->model.get());      \
  IDX_CHECK(m, "not an s2mel context")

size_t idxtts_s2mel_cond_workspace_bytes(const idxtts_ctx* ctx, int B, int M, int Tg) {
  if (!ctx || !ctx->finalized || B <= 0 || M <= 0 || Tg <= 0) return 0;
  auto* m = dynamic_cast<const S2MelModel*>(ctx->model.get());
  return m ? m->cond_workspace_bytes(B, M, Tg) : 0;
}

int idxtts_s2mel_prepare_cond(idxtts_ctx* ctx, const float* latent, const long long* codes, const int* code_lens,
                              const int* target_lens, int B, int M, int Tg, float* cond_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  API_BEGIN
  S2MEL_MODEL(ctx);
  return m->prepare_cond(latent, codes, code_lens, target_lens, B, M, Tg, cond_out, workspace, workspace_bytes,
                         static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_s2mel_regulate(idxtts_ctx* ctx, const float* S, const int* in_lens, const int* target_lens, int B, int M, int Tg, float* cond_out,
                          void* workspace, size_t workspace_bytes, void* stream) {
  API_BEGIN
  S2MEL_MODEL(ctx);
  return m->regulate(S, in_lens, target_lens, B, M, Tg, cond_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

size_t idxtts_s2mel_cfm_workspace_bytes(const idxtts_ctx* ctx, int B, int T, int n_steps) {
  if (!ctx || !ctx->finalized || B <= 0 || T <= 0 || n_steps <= 0) return 0;
  auto* m = dynamic_cast<const S2MelModel*>(ctx->model.get());
  return m ? m->cfm_workspace_bytes(B, T, n_steps) : 0;
}

int idxtts_s2mel_cfm(idxtts_ctx* ctx, const float* mu, const int* x_lens, const float* prompt, const int* prompt_lens, int Tp_max,
                     const float* style, const float* z, const float* t_emb, const float* dt, int n_steps, float cfg_rate,
                     float* out, int B, int T, void* workspace, size_t workspace_bytes, void* stream) {
  API_BEGIN
  S2MEL_MODEL(ctx);
  return m->cfm(mu, x_lens, prompt, prompt_lens, Tp_max, style, z, t_emb, dt, n_steps, cfg_rate, out, B, T, workspace,
                workspace_bytes, static_cast<hipStream_t>(stream));
  API_END
}

int idxtts_s2mel_estimator(idxtts_ctx* ctx, const float* x, const float* prompt, const int* prompt_lens, int Tp_max, const int* x_lens,
                           const float* t_emb, const float* style, const float* mu, float* out, int B, int T, void* workspace,
                           size_t workspace_bytes, void* stream) {
  API_BEGIN
  S2MEL_MODEL(ctx);
  return m->estimator(x, prompt, prompt_lens, Tp_max, x_lens, t_emb, style, mu, out, B, T, workspace, workspace_bytes,
                      static_cast<hipStream_t>(stream));
  API_END
}


int idxtts_set_gemm_mode(int mode) {
  set_gemm_mode(mode);
  return 0;
}

int idxtts_get_gemm_mode(void) { return get_gemm_mode(); }

int idxtts_set_decode_geometry(int narrow) {
  set_decode_geometry(narrow);
  return 0;
}

int idxtts_get_decode_geometry(void) { return get_decode_geometry(); }

int idxtts_release_stream(void* stream) {
  API_BEGIN
  hipStream_t st = static_cast<hipStream_t>(stream);
  return gemm_release_stream_scratch(st) || gemm_tn_release_stream_scratch(st) || s2mel_release_stream(st);
  API_END
}

int idxtts_set_decode_plane_rows(int min_rows) {
  API_BEGIN
  IDX_CHECK(min_rows == 0 || (min_rows >= 5 && min_rows <= 65), "0 (default), 5..64 or 65 (off)");
  set_decode_plane_rows(min_rows);
  return 0;
  API_END
}

int idxtts_get_decode_plane_rows(void) { return get_decode_plane_rows(); }

int idxtts_s2mel_set_overlap(int on) {
  set_s2mel_overlap(on);
  return 0;
}

int idxtts_s2mel_get_overlap(void) { return get_s2mel_overlap(); }

}  // extern "C"
