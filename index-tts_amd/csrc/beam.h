// Device side of beam search / beam-sample for the GPT decode loop (HF `_beam_search` + BeamSearchScorer, restated in
// oracle/gpt.py::generate_beam).  Everything of a step that depends on data -- candidate selection, hypothesis bookkeeping,
// beam re-indexing of the token history, the repetition-penalty masks and the KV cache -- stays on the device, so a step has
// no host round trip and can be replayed from a hipGraph like the greedy step.
#pragma once
#include "decode.h"

namespace idxtts {

constexpr int BEAM_MAX = 8;     // num_beams <= 8 (the reference default is 3)

struct BeamState {
  // per decode row r = b * nb + j
  const float* logits = nullptr;      // [R][V] raw lm_head output of this step
  float* proc = nullptr;              // [R][V] processed scores + running beam score (scratch)
  unsigned char* seen = nullptr;      // [R][V] ids present in the row's input_ids (repetition penalty)
  float* beam_scores = nullptr;       // [R]
  int* next_tok = nullptr;            // [R] chosen by select, consumed by reorder
  int* beam_idx = nullptr;            // [R] source row of each new beam (global row index)
  int* seq = nullptr; int seq_ld = 0; // [R][seq_ld] generated tokens of each live beam
  int* cur_tok = nullptr;             // [R] token fed to the next decode step
  // finished hypotheses per utterance: up to nb + 1 while one is being inserted (BeamHypotheses.add)
  double* hyp_score = nullptr;        // [B][BEAM_MAX + 1] in list (insertion) order
  int* hyp_len = nullptr;             // [B][BEAM_MAX + 1]
  int* hyp_slot = nullptr;            // [B][BEAM_MAX + 1] physical row of hyp_seq holding that hypothesis
  int* hyp_seq = nullptr;             // [B][BEAM_MAX + 1][seq_ld]
  int* hyp_n = nullptr;               // [B]
  double* hyp_worst = nullptr;        // [B]
  int* done = nullptr;                // [B]
  const DecodeState* st = nullptr;
  const float* exp_noise = nullptr;   // [steps][B][nb * V] Exp(1) draws (do_sample), or null: exp1_draw(seed, ...)
  unsigned long long seed = 0;
  void* kcache = nullptr; void* vcache = nullptr;     // [L][R][H][G][Smax] / [L][R][H][Smax][G] 16-byte granules
  int kv_gran = 16;                                   // G: granules per key (16 = fp32 cache, 8 = bf16 cache; decode.h)
  int B = 0, nb = 0, V = 0, stop_token = 0, L = 0, H = 0, Smax = 0, prompt_len = 0;   // prompt_len: KV positions shared by all beams (P + 1)
  int do_sample = 0, top_k = 0, early_stopping = 0;
  float penalty = 1.0f, temperature = 1.0f, top_p = 1.0f;
  double length_penalty = 0.0;
};

// log_softmax -> repetition penalty -> (sampling: temperature, top-k, top-p with min_tokens_to_keep = 2) -> + beam score
int beam_scores_forward(const BeamState& s, hipStream_t st);
// 2 * nb candidates per utterance (multinomial without replacement == top-(2 nb) of probs / q, or plain top-k), sorted by score;
// BeamSearchScorer.process: finished hypotheses, next beams, done flags
int beam_select_forward(const BeamState& s, hipStream_t st);
// re-index token history, seen masks, KV cache rows (generated positions only) by beam_idx; append the chosen tokens
int beam_reorder_forward(const BeamState& s, hipStream_t st);

}  // namespace idxtts
