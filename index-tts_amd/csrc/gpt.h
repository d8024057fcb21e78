#pragma once
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/idxtts.h"
#include "attention.h"
#include "beam.h"
#include "ctx.h"
#include "decode.h"
#include "gemm.h"
#include "gemv16.h"
#include "gemv_pl.h"
#include "norm.h"
#include "prof.h"

namespace idxtts {

struct GPTLayer {
  const float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  LinearWeights attn_l, proj_l, fc_l, fc2_l;     // MFMA-32x32 packed (prefill / latent pass)
  Gemv16Weights attn_g, proj_g, fc_g, fc2_g;     // stream-order packed (decode); attn_g / fc_g carry diag(ln_g) folded in
  Gemv32Weights attn_p, proj_p, fc_p, fc2_p;     // compact formats only: the bf16-MFMA order of the plane GEMV (gemv_pl.h), 5..64 rows
  const float *attn_u = nullptr, *attn_c = nullptr;   // folded LayerNorm 1: colsum(diag(g) W), b . W + bias
  const float *fc_u = nullptr, *fc_c = nullptr;       // folded LayerNorm 2
};

struct GPTModel : ModelBase {
  idxtts_gpt_config cfg;
  std::vector<GPTLayer> layers;
  const float *lnf_g = nullptr, *lnf_b = nullptr, *fn_g = nullptr, *fn_b = nullptr;
  Gemv16Weights head_g;
  Gemv32Weights head_p;
  const float* head_b = nullptr;
  const float *mel_emb = nullptr, *text_emb = nullptr, *mel_pos = nullptr, *text_pos = nullptr;
  int weight_fmt = WFMT_F32;      // storage format of the decode weight streams (quantize_weights)
  int kv_fmt = 0;                 // KV cache of the cached generation: 0 = fp32, 1 = bf16 (keys / values rounded when produced; decode.h)
  std::atomic<int> generating{0}; // generate() / generate_beam() calls in flight: idxtts_gpt_set_kv_format refuses to switch under them
  struct GenScope { GPTModel* m; explicit GenScope(GPTModel* mm) : m(mm) { m->generating.fetch_add(1); } ~GenScope() { m->generating.fetch_sub(1); } };
  size_t kv_layer_bytes(int B, int Smax) const { return (size_t)B * cfg.heads * Smax * 64 * (kv_fmt ? 2 : 4); }
  hipStream_t own_stream = nullptr;
  static constexpr int OOB_SLOTS = 64;
  int* oob_flag = nullptr;        // device ints (one per embed() call in flight): set by the embedding gather when an index exceeds its table
  std::atomic<unsigned> oob_next{0};
  ~GPTModel() override {
    for (GraphSlot& g : graph_cache) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }

  struct Buffers {
    float *x, *h, *qkv, *att, *ff;            // [B*S][..] prefill / latent activations
    char *kcache, *vcache; int Smax;          // [L][B][H] x (Smax x 64 elements) each; fp32 or bf16 elements (kv_fmt, decode.h)
    float *xd, *hd, *attd, *ffd;              // decode residual / final-normed / attention output / mlp hidden: A-fragment images
    // plane-GEMV decode step (gemv_pl.h; compact weight streams, more than 4 rows): every activation a plain fp32 row-major matrix,
    // per-16-column row statistics for the folded LayerNorms, K-part partial sums and arrival counters
    float *xrow, *hrow, *attrow, *ffrow, *stats, *pl_slab; unsigned* pl_cnt;
    float *qkvd, *logits, *slab;              // [B][3d], [B][V] row-major; [<=8][B][d] K-split partial sums of mlp.c_proj
    size_t frag_off, frag_bytes;              // the fragment-image region (zeroed once per generate: padding rows stay 0)
    unsigned char* seen; int *finished, *cur_tok, *kstart;
    float* attn_part; unsigned* attn_cnt;     // key-split decode attention: [B][H][<=16][66] partials, [B][H] counters
    unsigned* ksb_cnt;                        // [d/16] arrival counters of the fused K-split mlp.c_proj (zeroed per generate)
    DecodeState* state;
    long long* codes;                         // [B][max_new] generated codes of the call in flight (copied to the caller's tensor at the end)
    size_t bytes;
  };
  // Instantiated decode-step graphs of greedy generations, keyed by everything the captured launches depend on (workspace
  // address and carve, batch, penalty): a server replaying the same shapes on the same stream re-captures nothing.
  struct GraphSlot {
    void* ws = nullptr; size_t ws_bytes = 0; int B = 0, S = 0, max_new = 0; float penalty = 0.0f; int kv16 = 0, geom = 0;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long stamp = 0; bool in_use = false;
  };
  std::vector<GraphSlot> graph_cache;
  std::mutex graph_mu;
  unsigned long graph_stamp = 0;
  static constexpr size_t GRAPH_CACHE_MAX = 8;

  explicit GPTModel(const idxtts_gpt_config& c);
  bool accepts(const std::string& name) const override;
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  // Host transform of the staged tensors (before finalize) into the model a compact decode format can hold exactly:
  //   (ln_g, ln_b, W, b) of c_attn / c_fc -> (1, 0, Q(diag(ln_g) W), ln_b . W + b);  c_proj, mlp.c_proj, mel_head -> Q(W).
  // Every consumer (prefill, latent pass, decode) is then packed from these tensors, so they all run the SAME model.
  int quantize_weights(std::map<std::string, HostTensor>& t, int fmt);
  Buffers carve(void* ws, int B, int S, int max_new) const;
  size_t workspace_bytes(int B, int S, int max_new) const;
  int layer_full(int li, const Buffers& w, int B, int S, const int* kstart, bool store_kv, hipStream_t st);
  int head_and_sample(const Buffers& w, int B, const float* x, int ldx, bool x_frag, float penalty, long long* codes, int codes_ld,
                      float* logits_out, hipStream_t st);
  int decode_step(const Buffers& w, int B, float penalty, long long* codes, int codes_ld, float* logits_base, hipStream_t st);
  int decode_step_pl(const Buffers& w, int B, float penalty, long long* codes, int codes_ld, float* logits_base, hipStream_t st);
  bool use_pl(int B) const;        // this many decode rows run on the plane GEMV
  bool fused_tail(int B) const;    // ... and the generation in flight is greedy: sample + embed + advance are one launch
  // (the sampling mode of the generation in flight is thread-local state in gpt.hip: generate() is re-entrant across host threads,
  //  each call with its own workspace and stream)
  int generate(const float* inputs_embeds, const int* pad_left_host, int B, int P, int max_new, float penalty, const idxtts_sampling* sampling, long long* codes,
               int* n_steps_out, float* logits_out, void* ws, size_t ws_bytes, int use_graph, hipStream_t st, const long long* forced = nullptr);
  // Beam search / beam-sample (HF _beam_search; the reference's default decoding mode, infer_v2.py:714-722): B utterances x
  // num_beams rows through the same decode step, selection / hypotheses / re-indexing on the device (beam.hip).
  size_t beam_workspace_bytes(int B, int nb, int S, int max_new) const;
  int generate_beam(const float* inputs_embeds, const int* pad_left_host, int B, int P, int max_new, float penalty, const idxtts_beam* beam,
                    long long* codes, int* n_steps_out, void* ws, size_t ws_bytes, int use_graph, hipStream_t st);
  int latent(const float* emb, const int* pad_left_host, int B, int S, int mel_start, int M, float* latent_out, void* ws, size_t ws_bytes,
             hipStream_t st);
  int embed(float* out, int rows, const int* text_ids, const int* text_pos_idx, const int* mel_ids, const int* mel_pos_idx,
            const float* extra, const int* extra_idx, hipStream_t st);
};

// beam state of the generation this host thread is running (null = greedy / sampling); read by head_and_sample
extern thread_local const BeamState* tl_beam;
// cached positions (incl. the token in flight) of the eager decode step being launched: the profiler's byte accounting of the decode
// attention (0 while a captured graph replays: the host does not know the position then)
extern thread_local int tl_prof_pos;

}  // namespace idxtts
