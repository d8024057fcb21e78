#pragma once
#include "common.h"

namespace idxtts {

struct DecodeState {   // device-resident per-generation scalars (replay-friendly)
  int pos;       // KV index of the token being processed (= keys already in the cache)
  int mel_pos;   // row of mel_pos_embedding for that token (reference quirk: 0, 2, 3, 4, ... model_v2.py:175-177)
  int step;      // column of `codes` the sampler writes
  unsigned arrive;   // workgroups of the fused sample + embed + advance launch that have finished (0 between launches)
};

struct DecodeAttnArgs {
  const float* qkv_part = nullptr; int parts = 0; int part_rows = 0;   // [parts][part_rows][3d] c_attn split-K slab
  const float* qkv_bias = nullptr;                                      // [3d]
  // this layer's cache.  kv16 = 0: fp32, K [B][H][16][Smax][4], V [B][H][Smax][64].  kv16 = 1: bf16 (rounded to nearest even when a
  // key / value is produced, the new token's own included), K [B][H][8][Smax][8], V [B][H][Smax][64]: 16-byte granules either way
  void* kcache = nullptr; void* vcache = nullptr; int kv16 = 0;
  float* out = nullptr;                                                 // [B][d] as A-fragment images (frag_index), or
  float* out_row = nullptr;                                             // as fp32 rows [B][d] (the plane GEMV's input, gemv_pl.h)
  const int* kstart = nullptr;                                          // [B] first valid key (left pad), or null
  const DecodeState* st = nullptr;
  int B = 0, H = 0, Smax = 0, d = 0;
  float scale = 0.125f;
  int pos_hint = 0;             // cached positions incl. this step as the HOST knows them (profiler accounting only; 0 under graph replay)
  // Key split for small batches (B * H workgroups cannot pull the KV stream through 256 CUs: a CU sustains ~25 GB/s):
  // nsplit > 1 workgroups per (utterance, head) each take a contiguous range of the keys and leave (max, sum, unnormalised
  // output); the last one to arrive merges them in split order (wait-free, decode_attn_nsplit() depends on B and H only).
  int nsplit = 1;
  float* part = nullptr;        // [B][H][nsplit][66]
  unsigned* cnt = nullptr;      // [B][H] arrival counters: 0 on entry, left at 0
};
int decode_attn_nsplit(int B, int H);
int decode_attn_forward(const DecodeAttnArgs& a, hipStream_t stream);

struct SampleArgs {
  const float* part = nullptr; int parts = 0; int part_rows = 0;   // [parts][part_rows][V] lm_head split-K slab
  const float* bias = nullptr;                                     // [V]
  float* logits_out = nullptr;                                     // optional [B][V] raw logits of this step
  unsigned char* seen = nullptr;                                   // [B][V] ids present in input_ids
  int* finished = nullptr;                                         // [B]
  long long* codes = nullptr; int codes_ld = 0;                    // [B][codes_ld]
  int* cur_tok = nullptr;                                          // [B]
  const DecodeState* st = nullptr;
  int B = 0, V = 0, stop_token = 0;
  float penalty = 1.0f;
  // teacher forcing (idxtts_gpt_generate_forced): `codes` still records the row's own argmax, but the token fed back (cur_tok, the
  // repetition-penalty set, the finished flag) is forced[b][step]
  const long long* forced = nullptr; int forced_ld = 0;
  // Fused tail of a greedy step (embed.x_row or embed.x_frag != null): the workgroup that picked row b's token also writes the NEXT
  // step's input x[b] = mel_emb[token] + mel_pos[st->mel_pos + 1] (model_v2.py:173-177) -- as the fp32 residual row
  // and the per-16-column row statistics of the first folded LayerNorm -- and the last workgroup to arrive advances the
  // step scalars (DecodeState::arrive counts them): sample + embed + advance in ONE launch.
  struct Embed {
    float* x_row = nullptr; float* x_stats = nullptr;      // plane-GEMV step: fp32 row + per-16-column statistics, or
    float* x_frag = nullptr;                               // fp32-MFMA GEMV step: the A-fragment image (frag_index)
    const float* mel_emb = nullptr; const float* mel_pos = nullptr; int d = 0;
    DecodeState* st_rw = nullptr;
  } embed;
};
int sample_greedy_forward(const SampleArgs& a, hipStream_t stream);

// Multinomial sampling of the next token (HF _sample, transformers_generation_utils.py:3222-3250, with the warpers of
// 1036-1044): repetition penalty -> / temperature -> top-k (ties at the threshold kept) -> top-p (ascending sort, softmax,
// cumulative <= 1 - top_p removed, largest always kept) -> softmax -> torch.multinomial(probs, 1), which IS
// argmax(probs / q) with q ~ Exp(1): the draw is an explicit input, exp_noise[step][b][V].
// mode SAMPLE_ACCEL restates the accel engine's Sampler instead (accel_engine.py:648-659): softmax(logits / T) divided by
// clamp_min(q, 1e-10), argmax; no penalty, no top-k / top-p.
enum SampleMode { SAMPLE_HF = 1, SAMPLE_ACCEL = 2 };
// The Exp(1) draw of element `idx` of a generation: the caller's tensor when it supplied one (reproduces torch's stream), else a
// counter-based generator -- splitmix64 of (seed, idx) -> u in (0, 1), 24 bits, never 0 or 1 -> -log(u) -- so that default calls need no
// [steps][B][V] noise tensor (0.8 GB for 16 utterances x 1500 steps; 2.4 GB with 3 beams).
__host__ __device__ static inline float exp1_draw(const float* noise, unsigned long long seed, size_t idx) {
  if (noise) return noise[idx];
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  // the open interval: u = 1 would make the draw 0 and probs / q infinite (or 0 / 0 for a filtered token) -- torch's exponential_
  // never returns 0 either
  const float u = ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
  return -logf(u);
}
struct SampleWarpArgs {
  SampleArgs base;                   // logits source, bookkeeping buffers, penalty (parts must be 1)
  int mode = SAMPLE_HF;
  float temperature = 1.0f;
  int top_k = 0;                     // 0 = off
  float top_p = 1.0f;                // >= 1 = off
  const float* exp_noise = nullptr;  // [steps][B][V], or null: draws come from exp1_draw(seed, ...)
  unsigned long long seed = 0;
};
int sample_warp_forward(const SampleWarpArgs& a, hipStream_t stream);

int advance_state(DecodeState* st, hipStream_t stream);
// qkv [B][S][3d] -> caches, positions [0, S).  kv16: the caches hold bf16 and the k / v columns of qkv are rounded IN PLACE, so the
// prefill attention that follows sees exactly the keys and values later steps read from the cache
int kv_store_prefill(float* qkv, void* kcache, void* vcache, int kv16, int B, int H, int S, int Smax, int d, hipStream_t stream);
constexpr int GATHER_MAX_TABLES = 5;
struct GatherArgs {
  float* out = nullptr; int ld_out = 0; int d = 0;
  const float* table[GATHER_MAX_TABLES] = {};   // [n_t][d]
  const int* idx[GATHER_MAX_TABLES] = {};       // [rows], -1 = skip
  int table_rows[GATHER_MAX_TABLES] = {};       // rows of each table (0 = not checked): an id >= table_rows is NOT read ...
  int* oob = nullptr;                           // ... and, when given, *oob is set to 1 + table index (device int, caller zeroes it)
};
int gather_sum_rows(const GatherArgs& a, int rows, hipStream_t stream);
int embed_step(float* x, int B, int d, const float* mel_emb, const float* mel_pos, const int* cur_tok, const DecodeState* st,
               hipStream_t stream);
// the same for the plane-GEMV path: x as fp32 rows [B][d] + the row statistics per 16 columns
int embed_step_pl(float* x_row, float* x_stats, int B, int d, const float* mel_emb, const float* mel_pos, const int* cur_tok,
                  const DecodeState* st, hipStream_t stream);

}  // namespace idxtts
