/* libidxtts_hip -- C ABI of the MI355X-native IndexTTS-2 hot path (gfx950 only).
 *
 * Drop-in boundary.  Every entry point takes plain device pointers, explicit
 * shapes and a hipStream_t (passed as void*), returns 0 on success / non-zero on error (message via
 * idxtts_last_error(), thread-local) and never throws.  No ownership transfer: the caller's allocator
 * (torch) owns inputs, outputs and workspaces; the library owns only an opaque idxtts_ctx holding the
 * kernel-layout (packed) weights.  One ctx per (process, device); calls on one ctx are serialised by
 * the caller (same rule as the reference object, infer_v2.py:304-310 / serve_tars.py:139-140).
 * All tensors are float32, contiguous, unless a parameter says otherwise.
 *
 * Reference interfaces replaced (file:line relative to grantjr1842/index-tts):
 *   idxtts_aa_act_fwd      <- pybind `anti_alias_activation_cuda.forward(input, up_ftr, down_ftr, alpha, beta)`
 *                             indextts/s2mel/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation.cpp:19-23,
 *                             `extern "C" fwd_cuda` anti_alias_activation_cuda.cu:212-246, used by
 *                             `FusedAntiAliasActivation.forward` activation1d.py:21-26
 *   idxtts_bigvgan_*       <- `BigVGAN.forward` indextts/s2mel/modules/bigvgan/bigvgan.py:360-386 as called at
 *                             indextts/infer_v2.py:860 (`self.bigvgan(vc_target.float())`)
 *   idxtts_conv1d_*        <- torch.nn.Conv1d / ConvTranspose1d call sites on the hot path
 *                             (bigvgan.py:136-139, 362, 367; wavenet.py:149, 161; length_regulator.py:51, 61)
 */
#ifndef IDXTTS_H
#define IDXTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idxtts_ctx idxtts_ctx;

#define IDXTTS_DTYPE_F32 0
#define IDXTTS_DTYPE_F16 1
#define IDXTTS_DTYPE_BF16 2

int idxtts_version(void);
/* Last error message of the calling thread ("" if none). */
const char* idxtts_last_error(void);

/* ---- fused anti-aliased SnakeBeta activation (reference: fwd_cuda, .cu:212-246) -------------------
 * out/in: [B][C][T] of element type `dtype` (IDXTTS_DTYPE_F32 / F16 / BF16: the dispatch of .cu:232-244; 16-bit inputs are
 * widened on load, the arithmetic is float32 throughout, the result is rounded once on store); up_filter/down_filter: 12 float32
 * taps; log_alpha/log_beta: float32 [C] (log-scale, exp'd in-kernel, .cu:86-89).  T == 0 is a no-op (.cu:193).  out must not alias in. */
int idxtts_aa_act_fwd(void* out, const void* in, const float* up_filter, const float* down_filter,
                      const float* log_alpha, const float* log_beta, int B, int C, int T, int dtype,
                      void* stream);

/* ---- stand-alone Conv1d on the fp32 matrix core (used by the tests and by s2mel) -------------------
 * weight: torch layout [Cout][Cin][K] (or [Cin][Cout][Kt] when transposed_stride > 1), host or device.
 * The returned handle owns the packed copy; free with idxtts_conv1d_destroy. */
typedef struct idxtts_conv1d idxtts_conv1d;
int idxtts_conv1d_create(const float* weight, const float* bias /* may be NULL */, int Cout, int Cin, int K,
                         int transposed_stride /* 1 = Conv1d, u>1 = ConvTranspose1d(k=2u, stride u, pad u/2) */,
                         idxtts_conv1d** out);
/* y[b][co][t] = scale*(bias + sum x[..t + k*dil - pad_left..] + residual) (+ y if accumulate).
 * pad_mode: 0 zero, 1 reflect.  x: [B][Cin][T]; y/residual: [B][Cout][T*transposed_stride]. */
int idxtts_conv1d_fwd(const idxtts_conv1d* conv, const float* x, float* y, const float* residual /* may be NULL */,
                      int B, int T, int dilation, int pad_left, int pad_mode, float scale, int accumulate,
                      void* stream);
int idxtts_conv1d_destroy(idxtts_conv1d* conv);

/* ---- generic weight hand-over for model contexts -------------------------------------------------
 * `name` is the reference module's state_dict key (e.g. "resblocks.3.convs1.0.weight"); data may be a
 * host or a device pointer (copied during the call).  Unknown keys are rejected; buffers the kernels
 * derive themselves ("*.filter") are accepted and cross-checked. */
int idxtts_ctx_load_tensor(idxtts_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim);
/* Pack everything loaded so far into kernel layouts on the current device; fails if a tensor is missing. */
int idxtts_ctx_finalize(idxtts_ctx* ctx);
/* Read a staged tensor back (host fp32, `capacity` floats) before finalize: after idxtts_gpt_quantize_weights this is the
 * exact model every kernel will run, under the reference's own keys (what a parity check feeds the reference / oracle). */
int idxtts_ctx_get_tensor(idxtts_ctx* ctx, const char* name, float* host_out, size_t capacity);
int idxtts_ctx_destroy(idxtts_ctx* ctx);
/* OCP fp8 e4m3fn value of a code / nearest-even saturating code of a value (host helpers of the fp8 weight format). */
float idxtts_fp8_e4m3_decode(unsigned char code);
unsigned char idxtts_fp8_e4m3_encode(float v);

/* ---- BigVGAN-v2 vocoder (reference: BigVGAN.forward, bigvgan.py:360-386) ------------------------- */
typedef struct idxtts_bigvgan_config {
  int num_mels;                   /* 80 */
  int upsample_initial_channel;   /* 1536 */
  int num_upsamples;              /* 6 */
  int upsample_rates[8];          /* 4,4,2,2,2,2 */
  int upsample_kernel_sizes[8];   /* 8,8,4,4,4,4 */
  int num_kernels;                /* 3 */
  int resblock_kernel_sizes[4];   /* 3,7,11 */
  int resblock_dilations[4][3];   /* 1,3,5 each */
} idxtts_bigvgan_config;

int idxtts_bigvgan_create(const idxtts_bigvgan_config* cfg, idxtts_ctx** out);
/* Bytes of scratch idxtts_bigvgan_fwd needs for a [B][num_mels][Tm] input. */
size_t idxtts_bigvgan_workspace_bytes(const idxtts_ctx* ctx, int B, int Tm);
/* mel: [B][num_mels][Tm] -> wav: [B][1][Tm*prod(upsample_rates)], clamped to [-1,1] (bigvgan.py:384).
 * If stage_out != NULL the activation tensor after up-sampling stage `stage_idx` (1-based, after the
 * resblock average, [B][C_i][T_i]) is also copied there (parity tests).  If clamp == 0 the final clamp
 * is skipped (tests compare the pre-clamp waveform so saturation cannot hide an error). */
int idxtts_bigvgan_fwd(idxtts_ctx* ctx, const float* mel, float* wav, int B, int Tm, void* workspace,
                       size_t workspace_bytes, int clamp, int stage_idx, float* stage_out, void* stream);
/* Ragged batch: mel_lengths = device int32 [B], valid frames of each row (<= Tm; mel must be zero beyond them).  Row b of the
 * result equals idxtts_bigvgan_fwd on mel[b, :, :mel_lengths[b]] alone -- every layer pads (zeros for the convolutions,
 * replicate for the anti-alias filters) at the row's OWN end, as the reference's per-utterance call does (infer_v2.py:860);
 * samples beyond mel_lengths[b] * prod(upsample_rates) are not meaningful. */
int idxtts_bigvgan_fwd_ragged(idxtts_ctx* ctx, const float* mel, const int* mel_lengths, float* wav, int B, int Tm, void* workspace,
                              size_t workspace_bytes, int clamp, void* stream);

/* ---- token-major building blocks (also used stand-alone by the tests) ------------------------------
 * idxtts_linear: Y[M][N] = act(X[M][K] W^T + bias) (+ residual) on the fp32 matrix core.
 * weight: [N][K] (torch nn.Linear) or, if weight_is_kn, [K][N] (HF Conv1D: model_v2.py:290 GPT2Model).
 * act: 0 none, 1 gelu_new, 2 silu, 3 swiglu (rows packed [32 gate | 32 linear] blocks; Y is [M][N/2]), 4 mish. */
typedef struct idxtts_linear idxtts_linear;
int idxtts_linear_create(const float* weight, const float* bias /* may be NULL */, int N, int K, int weight_is_kn,
                         idxtts_linear** out);
int idxtts_linear_fwd(const idxtts_linear* lin, const float* x, int ldx, float* y, int ldy, const float* residual /* may be NULL */,
                      int ldr, int M, int act, int bf16x3 /* 0: exact fp32 MFMA, 1: split-bf16 (3 bf16 MFMAs per product) */,
                      void* stream);
/* Arithmetic of the compute-bound passes of the model contexts (s2mel GEMMs and DiT attention, GPT latent pass, vocoder
 * convolutions): 0 = exact fp32 MFMA, 1 (default) = split-bf16: x*w ~= hi*hi' + hi*lo' + lo*hi' on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (relative product error ~2^-16; mel / waveform stay ~1e-5 from the fp32
 * reference).  The KV-cached greedy decode and its prefill are always exact fp32 (token indices are bit-exact in both modes). */
int idxtts_set_gemm_mode(int mode);
int idxtts_get_gemm_mode(void);
/* Geometry of the decode step's GEMVs (5..16 rows, bf16 / fp8 weight streams): 0 (default) = 1024-thread workgroups, the fastest form
 * for a decode that has the GPU to itself; 1 = 512-thread workgroups at <= 88 VGPRs, which fit into what ONE retiring workgroup of
 * an s2mel / vocoder kernel frees on a CU: 7 % slower alone, but a serving loop that decodes beside the acoustic stages of other
 * batches (indextts_amd/serving.py) gains 1.5 % (profiles/README.md "Round 3").  The two forms split K over a different number of
 * waves, so results differ in the last bits: choose once per process, before generating (cached decode graphs carry the choice). */
int idxtts_set_decode_geometry(int narrow);
int idxtts_get_decode_geometry(void);
/* From how many decode rows on (compact weight streams) the decode step runs on the plane GEMV (csrc/gemv_pl.hip: activations split
 * into three bf16 planes inside the kernel, bf16 MFMA with exact products, K split over waves and workgroups, one weight stream
 * for up to 64 rows -- the accel engine's one-graph-for-all-rows batching, accel/accel_engine.py:221-310) instead of the fp32-MFMA
 * GEMV.  Default 17: two or more 16-row tiles (merged requests, 16 utterances x 3 beams: 7 % faster at 48 rows); at <= 16 rows the
 * fp32-MFMA GEMV's single-round-trip launches win by 8 %.  5..64 selects a threshold, 65 turns the plane path off, 0 restores the
 * default.  Rows of 17..64-row decodes do not depend on the batch they are in; the two kernel families differ in summation order
 * (last bits), so choose once per process, before generating (cached decode graphs carry the choice). */
int idxtts_set_decode_plane_rows(int min_rows);
int idxtts_get_decode_plane_rows(void);
/* The CFM solver (idxtts_s2mel_cfm) can evaluate the conditional and the null half of its stacked batch (flow_matching.py:91-103,
 * one DiT.forward on 2B rows there) as two chains of launches on two streams, the null half a few kernels behind: same kernels
 * on the same rows, bit-identical results, 9 % less time for a solver that has the device to itself (one half's HBM-bound
 * epilogues beside the other's MFMA-bound loops).  on = 0 (default): one stacked 2B batch on the caller's stream, as the
 * reference evaluates it -- serving loops with several decode chains in flight should leave it off (profiles/README.md "Round 3"). */
int idxtts_s2mel_set_overlap(int on);
int idxtts_s2mel_get_overlap(void);
int idxtts_linear_destroy(idxtts_linear* lin);
/* The library keeps a little state per caller STREAM (the split-plane scratch of the LDS-DMA GEMM; the side stream and events of
 * idxtts_s2mel_set_overlap): a caller that retires a stream (a serving loop shutting down its worker threads) hands it back here,
 * before destroying the stream.  Stream-ordered, no device-wide synchronisation.  No reference counterpart (torch's caching
 * allocator plays this role there). */
int idxtts_release_stream(void* stream);
/* Multi-head attention, head_dim 64, softmax(q k^T * scale + mask) v without materialising the scores.
 * q/k/v/o are read in place: element (b, t, h, e) at base + b*batch_stride + t*token_stride + 64*h + e.
 * causal: key <= query.  kstart/kend: optional device int32 [B], keys outside [kstart, kend) are masked
 * (left padding of GPT prompts, model_v2.py:766-779; key padding of the DiT, diffusion_transformer.py:235-237). */
int idxtts_attention_fwd(const float* q, const float* k, const float* v, float* o, long q_batch_stride, int q_token_stride,
                         long kv_batch_stride, int kv_token_stride, long o_batch_stride, int o_token_stride, int B, int H,
                         int Sq, int Sk, int causal, const int* kstart, const int* kend, float scale, void* stream);
/* Same contract on the bf16 matrix core with split operands (x = hi + lo, three bf16 MFMAs per product, fp32
 * accumulation; relative error ~2^-16).  Used by the s2mel DiT and the GPT latent pass when the GEMM mode is
 * IDXTTS_GEMM_BF16X3; the KV-cache-building prefill always runs the exact-fp32 form above. */
int idxtts_attention_bf16x3_fwd(const float* q, const float* k, const float* v, float* o, long q_batch_stride, int q_token_stride,
                         long kv_batch_stride, int kv_token_stride, long o_batch_stride, int o_token_stride, int B, int H,
                         int Sq, int Sk, int causal, const int* kstart, const int* kend, float scale, void* stream);
/* Non-causal self-attention (head_dim 64) with the "relative_key" distance embedding of the w2v-bert-2.0 encoder the reference's prompt block
 * runs (indextts/infer_v2.py:633-638 get_emb -> transformers Wav2Vec2BertSelfAttention, position_embeddings_type="relative_key"):
 *   scores[i][j] = scale * (q_i . k_j + q_i . rel_key[clamp(j - i, -rel_left, rel_right) + rel_left]),   rel_key [rel_left + rel_right + 1][64]
 * (one table for every head, at most 96 rows), keys >= kend[b] masked.  q / k / v share one batch / token stride (a fused qkv buffer);
 * split_bf16 = 0: exact-fp32 MFMAs, 1: split-bf16 products. */
int idxtts_attention_relkey_fwd(const float* q, const float* k, const float* v, float* o, long batch_stride, int token_stride,
                                long o_batch_stride, int o_token_stride, int B, int H, int S, const int* kend, float scale,
                                const float* rel_key, int rel_left, int rel_right, int split_bf16, void* stream);
/* y = LayerNorm(x) * gamma + beta over the last dim (eps), rows of length d. */
int idxtts_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int M, int d, float eps, void* stream);

/* ---- GPT stage (reference: UnifiedVoice.inference_speech model_v2.py:796-895, .forward 673-723) -----
 * State-dict keys accepted by idxtts_ctx_load_tensor: gpt.h.{i}.{ln_1,ln_2}.{weight,bias},
 * gpt.h.{i}.attn.{c_attn,c_proj}.{weight,bias}, gpt.h.{i}.mlp.{c_fc,c_proj}.{weight,bias}, gpt.ln_f.*, final_norm.*,
 * mel_head.*, mel_embedding.weight, text_embedding.weight, mel_pos_embedding.emb.weight, text_pos_embedding.emb.weight,
 * speed_emb.weight (UnifiedVoice.state_dict(), model_v2.py:381-443). */
typedef struct idxtts_gpt_config {
  int model_dim, heads, layers;          /* 1280, 20, 24 */
  int number_mel_codes;                  /* 8194 */
  int number_text_tokens;                /* 12000 (table has +1 rows) */
  int start_mel_token, stop_mel_token;   /* 8192, 8193 */
  int mel_pos_len, text_pos_len;         /* 1818, 602 */
} idxtts_gpt_config;
int idxtts_gpt_create(const idxtts_gpt_config* cfg, idxtts_ctx** out);
/* Scratch for sequences of S tokens per row (prefill: P+1; latent pass: 34+L+2+M+2) and max_new_tokens of decode. */
/* Compact storage of the GPT's linear weights (call between the last idxtts_ctx_load_tensor and idxtts_ctx_finalize).
 * Replaces the reference's reduced-precision switch (`use_fp16` -> self.gpt.half(), infer_v2.py:109, 145-146) and is what
 * BASELINE.json configs[4] names ("fp8 GPT GEMMs"): the decode step is bound by the weight stream, so the format decides
 * the bytes per token; the arithmetic stays the fp32 MFMA.
 *   format 0: fp32 (default, no-op)   1: bf16 (nearest-even)   2: fp8 e4m3fn with a power-of-two scale per output channel
 * The staged tensors are rewritten IN PLACE into an ordinary fp32 GPT-2 state that the format holds exactly --
 *   c_attn / c_fc: (ln.weight, ln.bias, W, b) -> (1, 0, Q(diag(ln.weight) W), ln.bias . W + b);   c_proj, mlp.c_proj, mel_head: Q(W)
 * (mathematically the same network up to Q) -- and prefill, latent pass and decode are all packed from them, so the three
 * passes run one model and idxtts_ctx_get_tensor returns exactly that model for the reference / oracle to run. */
int idxtts_gpt_quantize_weights(idxtts_ctx* ctx, int format);
/* Storage of the KV cache of the cached generation (idxtts_gpt_generate / _sampled / _beam): 0 = fp32 (default), 1 = bf16.
 * The other half of the reference's reduced-precision switch: with `use_fp16` the whole GPT, its `past_key_values` included, is
 * half precision (infer_v2.py:145-146; model_v2.py:149-151).  Here a key / value is rounded to bf16 (nearest even) once, when it
 * is produced -- in the prefill (whose own attention then reads the rounded values) and in every decode step (the new token's own
 * k / v included) -- and every product and sum stays fp32; the decode attention streams half the bytes (the KV read is the largest
 * HBM stream of a batched decode step: 1.7 GB against 1.0 GB of bf16 weights at 16 utterances).  The latent pass has no cache and
 * is unchanged.  Call any time between generations on the context; idxtts_gpt_workspace_bytes follows the format. */
int idxtts_gpt_set_kv_format(idxtts_ctx* ctx, int format);
int idxtts_gpt_get_kv_format(const idxtts_ctx* ctx);
/* Greedy generations keep their instantiated decode-step hipGraph per (workspace address and size, B, prompt length, max_new_tokens,
 * penalty): the same shapes on the same workspace replay it without re-capturing (at most 8 are kept, least recently used first
 * out).  Returns how many are held (diagnostics / tests), -1 for a non-GPT context. */
int idxtts_gpt_graph_cache_entries(idxtts_ctx* ctx);
size_t idxtts_gpt_workspace_bytes(const idxtts_ctx* ctx, int B, int S, int max_new_tokens);
/* out[r] = text_emb[text_ids[r]] + text_pos[text_pos_idx[r]] + mel_emb[mel_ids[r]] + mel_pos[mel_pos_idx[r]] + extra[extra_idx[r]],
 * every term skipped where its index is < 0 (or its index array is NULL).  Index arrays are device int32 [rows].
 * Builds the [pad|cond|text] prompt of prepare_gpt_inputs (model_v2.py:749-779) and the latent-pass input (704-716). */
int idxtts_gpt_embed(idxtts_ctx* ctx, float* out, int rows, const int* text_ids, const int* text_pos_idx, const int* mel_ids,
                     const int* mel_pos_idx, const float* extra, const int* extra_idx, void* stream);
/* Greedy KV-cached generation = the accel_engine.generate plugin slot (model_v2.py:871-883) with the HF greedy
 * semantics of the fallback path (do_sample=False, num_beams=1; RepetitionPenalty over the fake prefix + generated ids;
 * finished rows emit stop_mel_token; stops when every row has stopped or at max_new_tokens).
 * inputs_embeds: [B][P][d] device (the `tts_embeddings` argument); pad_left: HOST int32 [B] = number of left-pad rows
 * (attention_mask zeros), may be NULL.  codes: device int64 [B][max_new_tokens]; *n_steps (host) = columns that are valid.
 * logits_out: optional device [max_new_tokens][B][V] raw (pre-penalty) logits per step (disables graph replay). */
int idxtts_gpt_generate(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                        float repetition_penalty, long long* codes, int* n_steps, float* logits_out, void* workspace,
                        size_t workspace_bytes, int use_graph, void* stream);
/* Parity instrument: the greedy loop above TEACHER-FORCED on forced_codes (device int64 [B][max_new_tokens]) -- at every step the
 * row's own argmax (after the repetition penalty) is written to `codes`, and forced_codes[b][step] is what continues the sequence
 * (input_ids of the next step, the penalty set, the finished flag: transformers_generation_utils.py:3252-3264 with next_tokens
 * replaced).  With logits_out it yields the logits of a GIVEN token sequence, i.e. what a test compares with the reference's logits on
 * the reference's own codes when two correct implementations may part at a near-tie of the argmax.  Eager launches, no graph.
 * *n_steps = steps run (stops early only when every forced row has emitted the stop token). */
int idxtts_gpt_generate_forced(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                               float repetition_penalty, const long long* forced_codes, long long* codes, int* n_steps, float* logits_out,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same generation loop with multinomial sampling instead of argmax (reference default do_sample=True,
 * infer_v2.py:714-722; HF _sample transformers_generation_utils.py:3196-3262 with the warpers of 1036-1044):
 *   mode 1: repetition penalty -> / temperature -> top-k -> top-p -> softmax -> torch.multinomial(probs, 1);
 *   mode 2: the accel engine's Sampler (accel/accel_engine.py:648-659): softmax(logits / temperature) / clamp_min(q, 1e-10), argmax.
 * torch.multinomial with one draw per row is argmax(probs / q), q ~ Exp(1) from ONE exponential_() call on a [B][V] tensor
 * per step: the caller supplies those draws (exp_noise, device fp32 [max_new_tokens][B][V]), so a seeded torch generator on
 * the host side reproduces the reference token for token.  Beam search (num_beams > 1) is not covered. */
typedef struct idxtts_sampling {
  int mode;              /* 0 greedy (as idxtts_gpt_generate), 1 HF multinomial with warpers, 2 accel-engine sampler */
  float temperature;     /* > 0 */
  int top_k;             /* 0 = off */
  float top_p;           /* >= 1 = off; < 1 needs 0 < top_k <= 1024 */
  const float* exp_noise;
  unsigned long long seed;  /* used when exp_noise is NULL: the draws come from a counter-based generator keyed by (seed, step, row, id) */
} idxtts_sampling;
int idxtts_gpt_generate_sampled(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                                float repetition_penalty, const idxtts_sampling* sampling, long long* codes, int* n_steps,
                                float* logits_out, void* workspace, size_t workspace_bytes, int use_graph, void* stream);
/* Beam search / beam-sample = what `IndexTTS2.infer` runs by default (infer_v2.py:714-722, 767: do_sample=True, num_beams=3,
 * top_p .8, top_k 30, temperature .8, repetition_penalty 10, length_penalty 0) through HF `_beam_search`
 * (transformers_generation_utils.py:3325-3516) + BeamSearchScorer (transformers_beam_search.py:123-420): log_softmax before
 * the processors, warpers with min_tokens_to_keep = 2, 2 * num_beams candidates per utterance drawn without replacement
 * (torch.multinomial == top-(2 num_beams) of probs / q, q ~ Exp(1): exp_noise, device fp32 [max_new_tokens][B][num_beams * V],
 * one exponential_() per step on the [B][num_beams * V] tensor) or taken as the plain top-k (do_sample = 0), hypotheses kept
 * per utterance, the best one returned.  codes: device int64 [B][max_new_tokens] = best hypothesis, the stop token appended when
 * it fits, stop-token padded; *n_steps = valid columns.  B * num_beams <= 64, 2 <= num_beams <= 8. */
typedef struct idxtts_beam {
  int num_beams;
  int do_sample;
  float temperature;
  int top_k;
  float top_p;
  float length_penalty;
  int early_stopping;        /* 0 = False (heuristic, the default), 1 = True */
  const float* exp_noise;
  unsigned long long seed;   /* used when exp_noise is NULL (see idxtts_sampling) */
} idxtts_beam;
size_t idxtts_gpt_beam_workspace_bytes(const idxtts_ctx* ctx, int B, int num_beams, int S, int max_new_tokens);
int idxtts_gpt_generate_beam(idxtts_ctx* ctx, const float* inputs_embeds, const int* pad_left, int B, int P, int max_new_tokens,
                             float repetition_penalty, const idxtts_beam* beam, long long* codes, int* n_steps, void* workspace,
                             size_t workspace_bytes, int use_graph, void* stream);
/* Latent pass: full causal forward over emb [B][S][d]; latent[b][i] = final_norm(ln_f(h[b][mel_start + i])), i < M.
 * pad_left: optional HOST int32 [B], leading rows of each sequence that are padding (masked as keys), so rows with
 * shorter texts can share a batch and still reproduce the reference's per-utterance (B=1) result. */
int idxtts_gpt_latent(idxtts_ctx* ctx, const float* emb, const int* pad_left, int B, int S, int mel_start, int M, float* latent,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- prompt-conditioning encoders (reference: UnifiedVoice.get_conditioning model_v2.py:627-663, get_emo_conditioning
 * 665-671, get_emovec / merge_emovec 897-910; ConformerEncoder conformer_encoder.py:284-520; PerceiverResampler
 * perceiver.py:193-317).  One context per encoder pair:
 *   emotion = 0: keys "conditioning_encoder.*" + "perceiver_encoder.*"            -> latents [B][num_latents][perceiver_dim]
 *   emotion = 1: keys "emo_conditioning_encoder.*" + "emo_perceiver_encoder.*" + "emovec_layer.*" + "emo_layer.*"
 *                                                                                 -> emotion vector [B][model_dim] (get_emovec)
 * The sinusoid buffer "<encoder>.embed.pos_enc.pe" [1][max_len][output_size] of the reference's state_dict is a required
 * tensor (embedding.py:45-53). */
typedef struct idxtts_cond_config {
  int input_size;                                   /* 1024 (w2v-bert features) */
  int output_size, linear_units, attention_heads, num_blocks, cnn_kernel;   /* 512, 2048 | 1024, 8 | 4, 6 | 4, 15 */
  int perceiver_dim, num_latents, perceiver_depth, perceiver_dim_head, perceiver_mult;   /* 1280 | 1024, 32 | 1, 2, 64, 2 */
  int emotion;
  int model_dim;                                    /* 1280: width of emovec_layer / emo_layer (emotion = 1) */
} idxtts_cond_config;
int idxtts_cond_create(const idxtts_cond_config* cfg, idxtts_ctx** out);
size_t idxtts_cond_workspace_bytes(const idxtts_ctx* ctx, int B, int T);
/* feats: device [B][T][input_size]; lengths: HOST int32 [B] valid frames per prompt, or NULL = all T (values > T are
 * clamped: the reference passes the feature width 1024 as "length", infer_v2.py:751-752, i.e. no padding). */
int idxtts_cond_forward(idxtts_ctx* ctx, const float* feats, const int* lengths, int B, int T, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);
/* out = base + alpha * (emo - base) over n floats (merge_emovec, model_v2.py:904-910). */
int idxtts_emovec_merge(float* out, const float* base, const float* emo, float alpha, size_t n, void* stream);

/* ---- semantic features of a prompt (reference: IndexTTS2.get_emb, infer_v2.py:381-408; build_semantic_model,
 * utils/maskgct_utils.py:87-93) -------------------------------------------------------------------------
 * The first `num_layers` conformer layers of w2v-bert-2.0 (third-party: transformers' Wav2Vec2BertModel, pinned 4.52.1 by the
 * reference; `hidden_states[17]` of `output_hidden_states=True` is the INPUT of layer 17, i.e. 17 layers run), then
 * (x - semantic_mean) / semantic_std.  State-dict keys: Wav2Vec2BertModel's own ("feature_projection.*",
 * "encoder.layers.{i}.*" for i < num_layers; later layers and "masked_spec_embed" are not consumed), plus the optional
 * vectors "semantic_mean" / "semantic_std" [hidden_size] (wav2vec2bert_stats.pt: mean, sqrt(var)). */
typedef struct idxtts_w2vbert_config {
  int input_dim, hidden_size, num_heads, intermediate_size;   /* 160, 1024, 16, 4096 */
  int num_layers;                                             /* 17 */
  int left_max, right_max, conv_kernel;                       /* 64, 8, 31 (relative_key distance embedding; causal depthwise conv) */
  float layer_norm_eps;                                       /* 1e-5 */
} idxtts_w2vbert_config;
int idxtts_w2vbert_create(const idxtts_w2vbert_config* cfg, idxtts_ctx** out);
size_t idxtts_w2vbert_workspace_bytes(const idxtts_ctx* ctx, int B, int T);
/* feats: device [B][T][input_dim] (SeamlessM4TFeatureExtractor's input_features); lengths: HOST int32 [B] valid frames
 * (attention_mask.sum(1)) or NULL = all T; out: device [B][T][hidden_size] (rows t >= lengths[b] are unspecified). */
int idxtts_w2vbert_forward(idxtts_ctx* ctx, const float* feats, const int* lengths, int B, int T, float* out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- semantic codec, `quantize` (reference: `_, S_ref = self.semantic_codec.quantize(spk_cond_emb)`, infer_v2.py:637; RepCodec,
 * utils/maskgct/models/codec/kmeans/repcodec_model.py:179-199; build_semantic_codec, utils/maskgct_utils.py:96-99) ------
 * State-dict keys: RepCodec's own, "encoder.*" (VocosBackbone + Linear) and "quantizer.quantizers.0.{in_project,out_project,
 * codebook}.*" with the weight-norm pairs folded to plain ".weight"; "decoder.*" is not consumed. */
typedef struct idxtts_repcodec_config {
  int hidden_size, codebook_size, codebook_dim;                   /* 1024, 8192, 8 */
  int vocos_dim, vocos_intermediate_dim, vocos_num_layers;        /* 384, 2048, 12 */
} idxtts_repcodec_config;
int idxtts_repcodec_create(const idxtts_repcodec_config* cfg, idxtts_ctx** out);
size_t idxtts_repcodec_workspace_bytes(const idxtts_ctx* ctx, int B, int T);
/* x: device [B][T][hidden_size] (the normalised w2v-bert features); indices: device int64 [B][T]; quantized: device
 * [B][T][hidden_size] (= quantize(x)[1], already transposed back to token-major: the S_ref the length regulator takes). */
int idxtts_repcodec_quantize(idxtts_ctx* ctx, const float* x, int B, int T, long long* indices, float* quantized, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- log-mel spectrogram of the prompt (reference: `ref_mel = self.mel_fn(audio_22k)`, infer_v2.py:291-301, 640;
 * mel_spectrogram, s2mel/modules/audio.py:45-83) -----------------------------------------------------------------------
 * Tensors of the context: "mel_basis" [num_mels][n_fft/2+1] (the reference's librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax))
 * and "window" [win_size] (torch.hann_window(win_size)). */
typedef struct idxtts_melspec_config {
  int n_fft, hop_size, win_size, num_mels;          /* 1024, 256, 1024, 80 */
} idxtts_melspec_config;
int idxtts_melspec_create(const idxtts_melspec_config* cfg, idxtts_ctx** out);
int idxtts_melspec_frames(const idxtts_ctx* ctx, int n_samples);       /* (n_samples + 2 * ((n_fft - hop) / 2) - n_fft) / hop + 1 */
size_t idxtts_melspec_workspace_bytes(const idxtts_ctx* ctx, int B, int n_samples);
/* audio: device [B][n_samples] in [-1, 1]; mel: device [B][num_mels][frames] = log(clamp(mel_basis @ |STFT|, 1e-5)). */
int idxtts_melspec_forward(idxtts_ctx* ctx, const float* audio, int B, int n_samples, float* mel, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- CAMPPlus speaker encoder: the global style vector of the prompt (reference: `style = self.campplus_model(feat.unsqueeze(0))`,
 * infer_v2.py:251-257, 641-647; CAMPPlus, s2mel/modules/campplus/DTDNN.py:62-140, layers.py) ------------------------------
 * State-dict keys: CAMPPlus's own ("head.*", "xvector.*": campplus_cn_common.bin), BatchNorm running statistics included
 * (inference mode; "num_batches_tracked" entries are ignored). */
typedef struct idxtts_campplus_config {
  int feat_dim, embedding_size;                        /* 80, 192 */
  int m_channels, growth_rate, bn_size, init_channels; /* 32, 32, 4, 128 */
  int num_blocks;                                      /* 3 */
  int block_layers[4], block_dilation[4];              /* {12, 24, 16}, {1, 2, 2} (kernel 3) */
} idxtts_campplus_config;
int idxtts_campplus_create(const idxtts_campplus_config* cfg, idxtts_ctx** out);
size_t idxtts_campplus_workspace_bytes(const idxtts_ctx* ctx, int T);
/* feat: device [B][T][feat_dim] (Kaldi fbank minus its mean over time); style: device [B][embedding_size]. */
int idxtts_campplus_forward(idxtts_ctx* ctx, const float* feat, int B, int T, float* style, void* workspace, size_t workspace_bytes, void* stream);

/* ---- s2mel stage (reference: infer_v2.py:835-856; MyModel commons.py:390-420) -------------------------
 * State-dict keys: "cfm.estimator.*", "length_regulator.*", "gpt_layer.{0,1,2}.*" (s2mel.pth['net'][...], weight-norm
 * layers folded to plain ".weight"), "semantic_codec.quantizer.quantizers.0.{codebook.weight,out_project.weight,out_project.bias}",
 * plus one constant table "rope_cache" [T][32][2] = precompute_freqs_cis (gpt_fast/model.py:336-345). */
typedef struct idxtts_s2mel_config {
  int hidden_dim, num_heads, depth;      /* DiT: 512, 8, 13 */
  int in_channels, content_dim, style_dim;   /* 80, 512, 192 */
  int wn_hidden, wn_layers, wn_kernel, wn_dilation_rate;   /* 512, 8, 5, 1 */
  int lr_channels, lr_in_channels, lr_num_convs;           /* 512, 1024, 4 */
  int gpt_dim; int gpt_layer_dims[3];                      /* 1280; 256,128,1024 */
  int codebook_size, codebook_dim, codec_hidden;           /* 8192, 8, 1024 */
  float norm_eps;                                          /* 1e-5 */
} idxtts_s2mel_config;
int idxtts_s2mel_create(const idxtts_s2mel_config* cfg, idxtts_ctx** out);
/* cond = length_regulator(vq2emb(codes) + gpt_layer(latent)) (infer_v2.py:835-849).  latent [B][M][gpt_dim], codes int64
 * [B][M] (device); code_lens / target_lens: HOST int32 [B] (target = floor(1.72 * code_len), infer_v2.py:844);
 * cond_out [B][Tg][lr_channels], rows >= target_lens[b] are zero.  Statistics (GroupNorm) are per utterance. */
size_t idxtts_s2mel_cond_workspace_bytes(const idxtts_ctx* ctx, int B, int M, int Tg);
int idxtts_s2mel_prepare_cond(idxtts_ctx* ctx, const float* latent, const long long* codes, const int* code_lens,
                              const int* target_lens, int B, int M, int Tg, float* cond_out, void* workspace,
                              size_t workspace_bytes, void* stream);
/* length_regulator(S, ylens) alone (InterpolateRegulator.forward, length_regulator.py:117-141) = the prompt-side call of
 * infer_v2.py:649-652 that turns S_ref into prompt_condition: S [B][M][lr_in_channels]; in_lens / target_lens HOST int32 [B];
 * cond_out [B][Tg][lr_channels], rows >= target_lens[b] zero.  Workspace: idxtts_s2mel_cond_workspace_bytes(ctx, B, M, Tg). */
int idxtts_s2mel_regulate(idxtts_ctx* ctx, const float* S, const int* in_lens, const int* target_lens, int B, int M, int Tg, float* cond_out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* CFM Euler solve with classifier-free guidance = cfm.inference (flow_matching.py:31-115) with the noise given:
 * mu [B][T][content_dim]; x_lens HOST [B]; prompt [B][in_channels][Tp_max] + prompt_lens HOST [B]; style [B][style_dim];
 * z [B][in_channels][T] (the randn of flow_matching.py:52); t_emb device [n_steps][256] sinusoidal timestep features and
 * dt HOST [n_steps] (both from torch.linspace as the reference does); out [B][in_channels][T] (caller slices off the prompt). */
size_t idxtts_s2mel_cfm_workspace_bytes(const idxtts_ctx* ctx, int B, int T, int n_steps);
int idxtts_s2mel_cfm(idxtts_ctx* ctx, const float* mu, const int* x_lens, const float* prompt, const int* prompt_lens, int Tp_max,
                     const float* style, const float* z, const float* t_emb, const float* dt, int n_steps, float cfg_rate,
                     float* out, int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* One evaluation of the CFM estimator = DiT.forward(x, prompt_x, x_lens, t, style, cond) (diffusion_transformer.py:186-257),
 * the function `cfm.inference` calls once per Euler step on the [cond | null] stack (flow_matching.py:96).  x [B][in_channels][T];
 * prompt [B][in_channels][Tp_max] with prompt_lens HOST [B] (prompt_x is zero beyond them); x_lens HOST [B]; t_emb device
 * [1][256] sinusoidal features of the (shared) timestep; style [B][style_dim]; mu [B][T][content_dim].
 * out: TOKEN-MAJOR [B][T][in_channels] (the reference returns [B][in_channels][T]).  Workspace: idxtts_s2mel_cfm_workspace_bytes(ctx, B, T, 1). */
int idxtts_s2mel_estimator(idxtts_ctx* ctx, const float* x, const float* prompt, const int* prompt_lens, int Tp_max, const int* x_lens,
                           const float* t_emb, const float* style, const float* mu, float* out, int B, int T, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- per-kernel timing for the benchmark's roofline report ------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on its own stream and the library
 * accumulates, per kernel family, the launch count, elapsed milliseconds and the ALGORITHMIC flops /
 * bytes of the launches (DESIGN.md "work units").  Enabling resets the accumulators.  Not for use
 * inside a timed region (event records serialise nothing but add host work per launch). */
int idxtts_profile_enable(int on);
int idxtts_profile_num_kernels(void);
const char* idxtts_profile_kernel_name(int index);
int idxtts_profile_read(int index, double* total_ms, double* flops, double* bytes, long* launches);
/* Average reading of an event pair around one launch of an EMPTY 256-workgroup kernel on `stream`: the fixed per-launch cost
 * of this timing method (subtract launches * overhead before comparing kernel families by time). */
int idxtts_profile_event_overhead(void* stream, int launches, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* IDXTTS_H */
