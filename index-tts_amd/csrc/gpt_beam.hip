// Beam search / beam-sample generation on the GPT decode step (host side; kernels in beam.hip).
//
// Reference: UnifiedVoice.inference_speech with num_beams > 1 (indextts/gpt/model_v2.py:835-892) = HF generate ->
// _beam_search (indextts/gpt/transformers_generation_utils.py:2226-2255, 3325-3516) with BeamSearchScorer
// (indextts/gpt/transformers_beam_search.py:123-420): the decoding mode IndexTTS2.infer runs by default
// (infer_v2.py:714-722, 767).  Rows r = b * num_beams + j (_expand_inputs_for_generation = repeat_interleave).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "gpt.h"

namespace idxtts {

namespace {

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

struct BeamBuffers {
  float *proc, *beam_scores;
  int *next_tok, *beam_idx, *seq, *hyp_len, *hyp_slot, *hyp_seq, *hyp_n, *done;
  double *hyp_score, *hyp_worst;
  size_t bytes;
};

BeamBuffers carve_beam(void* ws, int B, int nb, int V, int max_new) {
  const int R = B * nb;
  BeamBuffers b;
  Carver c(ws);
  b.proc = c.take<float>((size_t)R * V);
  b.beam_scores = c.take<float>(R);
  b.next_tok = c.take<int>(R);
  b.beam_idx = c.take<int>(R);
  b.seq = c.take<int>((size_t)R * max_new);
  b.hyp_score = c.take<double>((size_t)B * (BEAM_MAX + 1));
  b.hyp_worst = c.take<double>(B);
  b.hyp_len = c.take<int>((size_t)B * (BEAM_MAX + 1));
  b.hyp_slot = c.take<int>((size_t)B * (BEAM_MAX + 1));
  b.hyp_seq = c.take<int>((size_t)B * (BEAM_MAX + 1) * max_new);
  b.hyp_n = c.take<int>(B);
  b.done = c.take<int>(B);
  b.bytes = (c.off + 255) & ~(size_t)255;
  return b;
}

}  // namespace

size_t GPTModel::beam_workspace_bytes(int B, int nb, int S, int max_new) const {
  return workspace_bytes(B * nb, S, max_new) + carve_beam(nullptr, B, nb, cfg.number_mel_codes, max_new).bytes;
}

int GPTModel::generate_beam(const float* inputs_embeds, const int* pad_left_host, int B, int P, int max_new, float penalty,
                            const idxtts_beam* beam, long long* codes, int* n_steps_out, void* ws, size_t ws_bytes, int use_graph,
                            hipStream_t user_stream) {
  IDX_CHECK(inputs_embeds && codes && n_steps_out && beam, "null pointer");
  GenScope gen_scope(this);
  hipStream_t st = user_stream;
  if (user_stream == nullptr) {
    if (!own_stream) IDX_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    IDX_HIP(hipStreamSynchronize(user_stream));
    st = own_stream;
  }
  const int nb = beam->num_beams, R = B * nb;
  IDX_CHECK(nb >= 2 && nb <= BEAM_MAX, "2 <= num_beams <= 8");
  IDX_CHECK(B > 0 && R <= 64 && P > 0 && max_new > 0, "shape (B * num_beams <= 64)");
  IDX_CHECK(!beam->do_sample || (beam->temperature > 0.0f && beam->top_k >= 0 && beam->top_k <= 1024 && beam->top_p > 0.0f),
            "beam-sample needs a positive temperature, top_k <= 1024 and top_p > 0");
  // the nucleus is cut inside the top-k survivors (beam_scores_kernel stages at most 2048 of them): without a top-k the whole
  // vocabulary would survive and the threshold would come from whichever 2048 entries arrived first
  IDX_CHECK(!beam->do_sample || beam->top_p >= 1.0f || (beam->top_k > 0 && beam->top_k <= 1024), "top-p needs 0 < top_k <= 1024");
  IDX_CHECK(beam->early_stopping == 0 || beam->early_stopping == 1, "early_stopping must be 0 (False) or 1 (True)");
  const int d = cfg.model_dim, V = cfg.number_mel_codes, S = P + 1;
  IDX_CHECK(max_new + 1 < cfg.mel_pos_len, "max_new_tokens exceeds the mel position table");
  IDX_CHECK(ws && ws_bytes >= beam_workspace_bytes(B, nb, S, max_new), "workspace too small");
  const size_t base_bytes = workspace_bytes(R, S, max_new);
  Buffers w = carve(ws, R, S, max_new);
  BeamBuffers bb = carve_beam(static_cast<char*>(ws) + base_bytes, B, nb, V, max_new);

  // ---- per-call state ----
  std::vector<int> kstart(R, 0);
  for (int r = 0; r < R; ++r) {
    kstart[r] = pad_left_host ? pad_left_host[r / nb] : 0;
    IDX_CHECK(kstart[r] >= 0 && kstart[r] < P, "pad_left out of range");
  }
  IDX_HIP(hipMemcpyAsync(w.kstart, kstart.data(), R * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemsetAsync(w.finished, 0, R * sizeof(int), st));
  IDX_HIP(hipMemsetAsync(w.ksb_cnt, 0, (size_t)cdiv(d, 16) * sizeof(unsigned), st));
  IDX_HIP(hipMemsetAsync(w.attn_cnt, 0, (size_t)R * cfg.heads * sizeof(unsigned), st));
  IDX_HIP(hipMemsetAsync(static_cast<char*>(ws) + w.frag_off, 0, w.frag_bytes, st));
  std::vector<unsigned char> seen((size_t)R * V, 0);
  for (int r = 0; r < R; ++r) { seen[(size_t)r * V + 1] = 1; seen[(size_t)r * V + cfg.start_mel_token] = 1; }
  IDX_HIP(hipMemcpyAsync(w.seen, seen.data(), seen.size(), hipMemcpyHostToDevice, st));
  DecodeState s0{P, 1, 0, 0};
  IDX_HIP(hipMemcpyAsync(w.state, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
  // first beam 0, the others -1e9: only the first beam's tokens count in the first step (transformers_generation_utils.py:3420-3422)
  std::vector<float> bs0(R, -1e9f);
  for (int b = 0; b < B; ++b) bs0[b * nb] = 0.0f;
  IDX_HIP(hipMemcpyAsync(bb.beam_scores, bs0.data(), R * sizeof(float), hipMemcpyHostToDevice, st));
  std::vector<double> worst0(B, 1e9);
  IDX_HIP(hipMemcpyAsync(bb.hyp_worst, worst0.data(), B * sizeof(double), hipMemcpyHostToDevice, st));
  IDX_HIP(hipMemsetAsync(bb.hyp_n, 0, B * sizeof(int), st));
  IDX_HIP(hipMemsetAsync(bb.done, 0, B * sizeof(int), st));
  std::vector<int> tok(R, cfg.start_mel_token);
  IDX_HIP(hipMemcpyAsync(w.cur_tok, tok.data(), R * sizeof(int), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));   // host staging buffers go out of scope below

  BeamState bst;
  bst.logits = w.logits; bst.proc = bb.proc; bst.seen = w.seen; bst.beam_scores = bb.beam_scores; bst.next_tok = bb.next_tok;
  bst.beam_idx = bb.beam_idx; bst.seq = bb.seq; bst.seq_ld = max_new; bst.cur_tok = w.cur_tok;
  bst.hyp_score = bb.hyp_score; bst.hyp_len = bb.hyp_len; bst.hyp_slot = bb.hyp_slot; bst.hyp_seq = bb.hyp_seq; bst.hyp_n = bb.hyp_n;
  bst.hyp_worst = bb.hyp_worst; bst.done = bb.done; bst.st = w.state; bst.exp_noise = beam->exp_noise; bst.seed = beam->seed;
  bst.kcache = w.kcache; bst.vcache = w.vcache; bst.kv_gran = kv_fmt ? 8 : 16;
  bst.B = B; bst.nb = nb; bst.V = V; bst.stop_token = cfg.stop_mel_token; bst.L = cfg.layers; bst.H = cfg.heads; bst.Smax = w.Smax;
  bst.prompt_len = S;
  bst.do_sample = beam->do_sample ? 1 : 0; bst.top_k = beam->top_k; bst.early_stopping = beam->early_stopping;
  bst.penalty = penalty; bst.temperature = beam->temperature; bst.top_p = beam->top_p; bst.length_penalty = (double)beam->length_penalty;
  struct Guard {      // the decode step reads the beam state through a thread-local (head_and_sample)
    Guard(const BeamState* s) { tl_beam = s; }
    ~Guard() { tl_beam = nullptr; }
  } guard(&bst);

  // ---- prefill on all R rows (the beams of an utterance start identical): x = [inputs_embeds | mel_emb[start] + mel_pos[0]] ----
  for (int r = 0; r < R; ++r)
    IDX_HIP(hipMemcpy2DAsync(w.x + (size_t)r * S * d, (size_t)S * d * sizeof(float), inputs_embeds + (size_t)(r / nb) * P * d,
                             (size_t)P * d * sizeof(float), (size_t)P * d * sizeof(float), 1, hipMemcpyDeviceToDevice, st));
  {
    GatherArgs ga;
    ga.out = w.x + (size_t)P * d; ga.ld_out = S * d; ga.d = d;
    ga.table[0] = mel_emb; ga.idx[0] = w.cur_tok;
    ga.table[1] = mel_pos; ga.idx[1] = w.finished;          // zeros -> mel position 0
    if (gather_sum_rows(ga, R, st)) return 1;
  }
  for (int li = 0; li < cfg.layers; ++li)
    if (layer_full(li, w, R, S, w.kstart, true, st)) return 1;
  if (head_and_sample(w, R, w.x + (size_t)(S - 1) * d, S * d, false, penalty, nullptr, max_new, nullptr, st)) return 1;
  if (advance_state(w.state, st)) return 1;

  // ---- decode ----
  struct GraphGuard {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    ~GraphGuard() { if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
  } gg;
  const bool graph_ok = use_graph && !prof_enabled();
  int n_first = 1;
  if (graph_ok && max_new > 2) {
    if (decode_step(w, R, penalty, nullptr, max_new, nullptr, st)) return 1;
    n_first = 2;
    IDX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = decode_step(w, R, penalty, nullptr, max_new, nullptr, st);
    const hipError_t e = hipStreamEndCapture(st, &gg.graph);
    if (rc) return 1;
    IDX_HIP(e);
    IDX_HIP(hipGraphInstantiate(&gg.exec, gg.graph, nullptr, nullptr, 0));
  }
  std::vector<int> done(B, 0);
  int steps_done = std::min(n_first, max_new);
  for (int n = n_first; n < max_new; ++n) {
    if (gg.exec) IDX_HIP(hipGraphLaunch(gg.exec, st));
    else {
      tl_prof_pos = S + n;      // keys this step reads (profiler accounting of the decode attention)
      const int rc = decode_step(w, R, penalty, nullptr, max_new, nullptr, st);
      tl_prof_pos = 0;
      if (rc) return 1;
    }
    steps_done = n + 1;
    if ((n & 7) == 7 || n + 1 == max_new) {     // beam_scorer.is_done (every utterance finished)?
      IDX_HIP(hipMemcpyAsync(done.data(), bb.done, B * sizeof(int), hipMemcpyDeviceToHost, st));
      IDX_HIP(hipStreamSynchronize(st));
      bool all = true;
      for (int b = 0; b < B; ++b) all = all && done[b];
      if (all) break;
    }
  }

  // ---- BeamSearchScorer.finalize on the host (transformers_beam_search.py:320-414) ----
  std::vector<int> seq((size_t)R * max_new), hyp_len((size_t)B * (BEAM_MAX + 1)), hyp_slot(hyp_len.size()), hyp_n(B);
  std::vector<int> hyp_seq((size_t)B * (BEAM_MAX + 1) * max_new);
  std::vector<double> hyp_score(hyp_len.size());
  std::vector<float> bscore(R);
  IDX_HIP(hipMemcpyAsync(seq.data(), bb.seq, seq.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(hyp_len.data(), bb.hyp_len, hyp_len.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(hyp_slot.data(), bb.hyp_slot, hyp_slot.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(hyp_n.data(), bb.hyp_n, B * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(hyp_seq.data(), bb.hyp_seq, hyp_seq.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(hyp_score.data(), bb.hyp_score, hyp_score.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(bscore.data(), bb.beam_scores, R * sizeof(float), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipMemcpyAsync(done.data(), bb.done, B * sizeof(int), hipMemcpyDeviceToHost, st));
  IDX_HIP(hipStreamSynchronize(st));
  struct Hyp { double score; const int* toks; int len; };
  std::vector<Hyp> best(B);
  int longest = 0;
  const double lp = (double)beam->length_penalty;
  for (int b = 0; b < B; ++b) {
    std::vector<Hyp> list;
    for (int q = 0; q < hyp_n[b]; ++q) {
      const size_t o = (size_t)b * (BEAM_MAX + 1) + q;
      list.push_back(Hyp{hyp_score[o], &hyp_seq[((size_t)b * (BEAM_MAX + 1) + hyp_slot[o]) * max_new], hyp_len[o]});
    }
    if (!done[b]) {      // open beams become hypotheses (BeamHypotheses.add with its keep-the-best rule)
      double worst = 1e9;
      for (const Hyp& h : list) worst = std::min(worst, h.score);
      for (int j = 0; j < nb; ++j) {
        const int r = b * nb + j;
        const double score = (double)bscore[r] / std::pow((double)steps_done, lp);
        if ((int)list.size() < nb || score > worst) {
          list.push_back(Hyp{score, &seq[(size_t)r * max_new], steps_done});
          if ((int)list.size() > nb) {
            size_t wi = 0;
            for (size_t q = 1; q < list.size(); ++q) if (list[q].score < list[wi].score) wi = q;
            list.erase(list.begin() + wi);
            worst = list[0].score;
            for (const Hyp& h : list) worst = std::min(worst, h.score);
          } else {
            worst = std::min(worst, score);
          }
        }
      }
    }
    IDX_CHECK(!list.empty(), "no hypothesis");
    size_t bi = 0;       // sorted(..., key=score).pop(): the largest score, the LAST of equals
    for (size_t q = 1; q < list.size(); ++q) if (list[q].score >= list[bi].score) bi = q;
    best[b] = list[bi];
    longest = std::max(longest, best[b].len);
  }
  const int n_out = std::min(longest + 1, max_new);          // min(sent_lengths.max() + 1, max_length) - prompt
  std::vector<long long> out((size_t)B * max_new, cfg.stop_mel_token);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < best[b].len; ++t) out[(size_t)b * max_new + t] = best[b].toks[t];      // then eos if it fits: already the fill value
  IDX_HIP(hipMemcpyAsync(codes, out.data(), out.size() * sizeof(long long), hipMemcpyHostToDevice, st));
  IDX_HIP(hipStreamSynchronize(st));
  *n_steps_out = n_out;
  return 0;
}

}  // namespace idxtts
