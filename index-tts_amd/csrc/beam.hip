// Beam search / beam-sample step kernels (see beam.h).  Reference semantics: vendored GenerationMixin._beam_search
// (indextts/gpt/transformers_generation_utils.py:3325-3516), BeamSearchScorer.process and BeamHypotheses
// (indextts/gpt/transformers_beam_search.py:215-318, 930-1013), logits warpers with min_tokens_to_keep = 2 (1022-1043),
// cache re-indexing `_reorder_cache` (indextts/gpt/model_v2.py:227-240) -- the mode IndexTTS2.infer runs by default
// (infer_v2.py:714-722, 767: do_sample=True, num_beams=3).
//
// The reference re-indexes the WHOLE KV cache of all 24 layers with index_select every step (3 x the prompt + everything
// generated).  Here the beams of an utterance share their prompt rows by construction (identical inputs -> identical
// K/V), so only the generated positions move, in place, and not at all when the step kept every beam where it was.
#include <cmath>

#include "beam.h"
#include "prof.h"

namespace idxtts {

namespace {

constexpr int NPT = 16;          // vocabulary entries per thread of a 1024-thread block: V <= 16384
constexpr int CAP = 2048;        // survivors of the top-k filter handled by the top-p stage
constexpr int SEL_EPT = 32;      // beam_select: candidates per thread kept in registers (nb * V <= 32768), else the re-reading loop

__device__ __forceinline__ void block_argmax1024(float& v, int& i, float* rv, int* ri, int tid) {   // max value, smallest index on ties
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(i, off);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { rv[tid >> 6] = v; ri[tid >> 6] = i; }
  __syncthreads();
  v = rv[0]; i = ri[0];
  for (int w = 1; w < 16; ++w)
    if (rv[w] > v || (rv[w] == v && ri[w] < i)) { v = rv[w]; i = ri[w]; }
}

__device__ __forceinline__ float block_sum1024(float v, float* rv, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0) rv[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < 16; ++w) t += rv[w];
  return t;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void beam_scores_kernel(const BeamState p) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  __shared__ float sval[CAP];
  __shared__ int sidx[CAP];
  __shared__ float sorted_v[CAP];
  __shared__ int sorted_i[CAP];
  __shared__ int s_count, s_thr_i, s_digit, s_kk;
  __shared__ int hist[16][256];
  __shared__ float s_thr_v;
  const int r = blockIdx.x, tid = threadIdx.x, V = p.V;
  if (p.done[r / p.nb]) return;
  const float* lrow = p.logits + (size_t)r * V;
  const unsigned char* seen = p.seen + (size_t)r * V;
  float sc[NPT];
  float mx = -INFINITY; int mi = 0;
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int v = tid + 1024 * u;
    sc[u] = v < V ? lrow[v] : -INFINITY;
    if (sc[u] > mx) { mx = sc[u]; mi = v; }
  }
  block_argmax1024(mx, mi, rv, ri, tid);
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NPT; ++u) if (tid + 1024 * u < V) part += expf(sc[u] - mx);
  const float lse = logf(block_sum1024(part, rv, tid));
  // log_softmax (fp32, before the processors: transformers_generation_utils.py:3473-3477), repetition penalty, temperature
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int v = tid + 1024 * u;
    if (v < V) {
      float s = (sc[u] - mx) - lse;
      if (p.penalty != 1.0f && seen[v]) s = s < 0.f ? s * p.penalty : s / p.penalty;
      if (p.do_sample && p.temperature != 1.0f) s = s / p.temperature;
      sc[u] = s;
    }
  }
  if (p.do_sample && (p.top_k > 0 || p.top_p < 1.0f)) {
    const int k = p.top_k > 0 ? max(p.top_k, 2) : 0;         // TopKLogitsWarper: max(top_k, min_tokens_to_keep)
    if (k > 0 && k < V) {
      // the k-th largest score by a radix select over order-preserving 32-bit keys (four 8-bit digits, high to low): per-wave
      // histograms in LDS, one wave walks the 256 bins from the top -- 16 barriers in all instead of 2 k block-wide argmax rounds
      unsigned key[NPT];
#pragma unroll
      for (int u = 0; u < NPT; ++u) {
        const unsigned b = __float_as_uint(sc[u] + 0.0f);      // (+ 0: -0 and +0 are one value to `<`)
        key[u] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      }
      unsigned prefix = 0;
      int kk = k;
      const int wv = tid >> 6, ln = tid & 63;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int j = 0; j < 4; ++j) (&hist[0][0])[tid + 1024 * j] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPT; ++u)
          if (tid + 1024 * u < V && (pass == 0 || (key[u] >> (shift + 8)) == prefix)) atomicAdd(&hist[wv][(key[u] >> shift) & 255u], 1);
        __syncthreads();
        if (tid < 256) {
          int t = 0;
#pragma unroll
          for (int w = 0; w < 16; ++w) t += hist[w][tid];
          hist[0][tid] = t;
        }
        __syncthreads();
        if (wv == 0) {      // lane l owns bins 4 l .. 4 l + 3; `above` = the count in every bin of a higher lane
          int c[4], mine = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) { c[j] = hist[0][4 * ln + j]; mine += c[j]; }
          int incl = mine;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_down(incl, off);
            if (ln + off < 64) incl += o;
          }
          int a = incl - mine;
          if (a < kk && kk <= incl) {
#pragma unroll
            for (int j = 3; j >= 0; --j) {
              if (a + c[j] >= kk) { s_digit = 4 * ln + j; s_kk = kk - a; break; }
              a += c[j];
            }
          }
        }
        __syncthreads();
        prefix = (prefix << 8) | (unsigned)s_digit;
        kk = s_kk;
      }
      const unsigned kb = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
      const float kth = __uint_as_float(kb);
#pragma unroll
      for (int u = 0; u < NPT; ++u) if (sc[u] < kth) sc[u] = -INFINITY;
    }
    if (p.top_p < 1.0f) {
      if (tid == 0) s_count = 0;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NPT; ++u) {
        const int v = tid + 1024 * u;
        if (v < V && sc[u] > -INFINITY) {
          const int slot = atomicAdd(&s_count, 1);
          if (slot < CAP) { sval[slot] = sc[u]; sidx[slot] = v; }
        }
      }
      __syncthreads();
      const int n = min(s_count, CAP);
      for (int e = tid; e < n; e += 1024) {        // rank sort, ascending by (value, index)
        const float ve = sval[e]; const int ie = sidx[e];
        int rank = 0;
        for (int o = 0; o < n; ++o) rank += (sval[o] < ve || (sval[o] == ve && sidx[o] < ie)) ? 1 : 0;
        sorted_v[rank] = ve; sorted_i[rank] = ie;
      }
      __syncthreads();
      if (tid == 0) {      // TopPLogitsWarper, min_tokens_to_keep = 2: remove while cum <= 1 - top_p, never the last two
        int keep_from = 0;
        const float top = sorted_v[n - 1];
        float tot = 0.f;
        for (int e = 0; e < n; ++e) tot += expf(sorted_v[e] - top);
        const float thr = (float)(1.0 - (double)p.top_p);
        float cum = 0.f;
        for (int e = 0; e < n - 2; ++e) {
          cum += expf(sorted_v[e] - top) / tot;
          if (cum <= thr) keep_from = e + 1; else break;
        }
        s_thr_v = sorted_v[keep_from]; s_thr_i = sorted_i[keep_from];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NPT; ++u) {
        const int v = tid + 1024 * u;
        if (sc[u] < s_thr_v || (sc[u] == s_thr_v && v < s_thr_i)) sc[u] = -INFINITY;
      }
    }
  }
  const float bs = p.beam_scores[r];
  float* out = p.proc + (size_t)r * V;
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int v = tid + 1024 * u;
    if (v < V) out[v] = sc[u] + bs;
  }
}

int beam_scores_forward(const BeamState& s, hipStream_t st) {
  IDX_CHECK(s.logits && s.proc && s.seen && s.beam_scores && s.done, "null pointer");
  IDX_CHECK(s.V > 0 && s.V <= 1024 * NPT && s.nb >= 2 && s.nb <= BEAM_MAX, "beam shape");
  static const int cat = prof_register("beam_scores_kernel");
  ProfScope prof(cat, st, 0.0, 8.0 * s.B * s.nb * (double)s.V);
  hipLaunchKernelGGL(beam_scores_kernel, dim3(s.B * s.nb), dim3(1024), 0, st, s);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void beam_select_kernel(const BeamState p) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  __shared__ int cand_i[2 * BEAM_MAX];
  __shared__ float cand_s[2 * BEAM_MAX];
  __shared__ int cp_slot[BEAM_MAX], cp_src[BEAM_MAX], n_cp;
  const int b = blockIdx.x, tid = threadIdx.x, nb = p.nb, V = p.V, N = nb * V;
  const int n = p.st->step;                   // tokens each live beam holds before this step
  if (p.done[b]) {                            // BeamSearchScorer.process pads a finished utterance
    if (tid < nb) { p.next_tok[b * nb + tid] = p.stop_token; p.beam_idx[b * nb + tid] = b * nb + tid; p.beam_scores[b * nb + tid] = 0.f; }
    return;
  }
  const float* base = p.proc + (size_t)b * N;
  const size_t nbase = ((size_t)n * p.B + b) * N;
  const int n_keep = 2 * nb;
  // multinomial without replacement = the n_keep largest probs / q (ATen: q ~ Exp(1) once per element, then topk); plain
  // top-k of the scores when not sampling.  Ties (only among zero-probability fillers) go to the smaller index.
  if (N <= 1024 * SEL_EPT) {
    // every element's key is computed ONCE and stays in registers (the exponential, the division and the draw are the cost of a
    // round); a round is then a masked scan of SEL_EPT registers and one block-wide argmax
    float ks[SEL_EPT];
    float mx = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e) {
      const int i = tid + 1024 * e;
      ks[e] = i < N ? base[i] : -INFINITY;
      if (i < N && (ks[e] > mx || (ks[e] == mx && i < mi))) { mx = ks[e]; mi = i; }
    }
    block_argmax1024(mx, mi, rv, ri, tid);
    if (p.do_sample) {
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < SEL_EPT; ++e) if (tid + 1024 * e < N) part += expf(ks[e] - mx);
      const float tot = block_sum1024(part, rv, tid);
#pragma unroll
      for (int e = 0; e < SEL_EPT; ++e) {
        const int i = tid + 1024 * e;
        if (i < N) ks[e] = (expf(ks[e] - mx) / tot) / exp1_draw(p.exp_noise, p.seed, nbase + i);
      }
    }
    unsigned taken = 0;
    for (int c = 0; c < n_keep; ++c) {
      float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < SEL_EPT; ++e) {
        const int i = tid + 1024 * e;
        if (i < N && !((taken >> e) & 1u) && (ks[e] > best || (ks[e] == best && i < bi))) { best = ks[e]; bi = i; }
      }
      block_argmax1024(best, bi, rv, ri, tid);
      if (bi < N && (bi & 1023) == tid) taken |= 1u << (bi >> 10);
      if (tid == 0) { cand_i[c] = bi; cand_s[c] = base[bi]; }
    }
    __syncthreads();
  } else {
    float mx = -INFINITY; int mi = 0x7fffffff;
    for (int i = tid; i < N; i += 1024) { const float x = base[i]; if (x > mx || (x == mx && i < mi)) { mx = x; mi = i; } }
    block_argmax1024(mx, mi, rv, ri, tid);
    float tot = 1.0f;
    if (p.do_sample) {
      float part = 0.f;
      for (int i = tid; i < N; i += 1024) part += expf(base[i] - mx);
      tot = block_sum1024(part, rv, tid);
    }
    for (int c = 0; c < n_keep; ++c) {
      float best = -INFINITY; int bi = 0x7fffffff;
      for (int i = tid; i < N; i += 1024) {
        bool taken = false;
        for (int o = 0; o < c; ++o) taken = taken || cand_i[o] == i;
        if (taken) continue;
        const float x = base[i];
        const float key = p.do_sample ? (expf(x - mx) / tot) / exp1_draw(p.exp_noise, p.seed, nbase + i) : x;
        if (key > best || (key == best && i < bi)) { best = key; bi = i; }
      }
      block_argmax1024(best, bi, rv, ri, tid);
      if (tid == 0) { cand_i[c] = bi; cand_s[c] = base[bi]; }
      __syncthreads();
    }
  }
  if (tid == 0) {
    // candidates by score, descending (transformers_generation_utils.py:3511-3513; stable)
    for (int a = 1; a < n_keep; ++a) {
      const int ci = cand_i[a]; const float cs = cand_s[a];
      int o = a - 1;
      while (o >= 0 && cand_s[o] < cs) { cand_i[o + 1] = cand_i[o]; cand_s[o + 1] = cand_s[o]; --o; }
      cand_i[o + 1] = ci; cand_s[o + 1] = cs;
    }
    double* hs = p.hyp_score + (size_t)b * (BEAM_MAX + 1);
    int* hl = p.hyp_len + (size_t)b * (BEAM_MAX + 1);
    int* hslot = p.hyp_slot + (size_t)b * (BEAM_MAX + 1);
    int hn = p.hyp_n[b];
    double worst = p.hyp_worst[b];
    const int gen_len = n + 1;                                  // cur_len - decoder_prompt_len
    const double denom = pow((double)gen_len, p.length_penalty);
    int slot = 0, ncp = 0;
    for (int rank = 0; rank < n_keep && slot < nb; ++rank) {
      const int tok = cand_i[rank] % V, src = b * nb + cand_i[rank] / V;
      const float sc = cand_s[rank];
      if (tok == p.stop_token) {
        if (rank >= nb) continue;                               // not among the top num_beams: dropped
        const double score = (double)sc / denom;                // BeamHypotheses.add
        if (hn < nb || score > worst) {
          bool used[BEAM_MAX + 1];
          for (int q = 0; q <= nb; ++q) used[q] = false;
          for (int q = 0; q < hn; ++q) used[hslot[q]] = true;
          int phys = 0;
          while (used[phys]) ++phys;
          hs[hn] = score; hl[hn] = n; hslot[hn] = phys; ++hn;
          cp_slot[ncp] = phys; cp_src[ncp] = src; ++ncp;
          if (hn > nb) {                                        // drop the worst (smallest score, then earliest)
            int wi = 0;
            for (int q = 1; q < hn; ++q) if (hs[q] < hs[wi]) wi = q;
            for (int q = wi; q + 1 < hn; ++q) { hs[q] = hs[q + 1]; hl[q] = hl[q + 1]; hslot[q] = hslot[q + 1]; }
            --hn;
            worst = hs[0];
            for (int q = 1; q < hn; ++q) worst = fmin(worst, hs[q]);
          } else {
            worst = fmin(score, worst);
          }
        }
      } else {
        p.beam_scores[b * nb + slot] = sc; p.next_tok[b * nb + slot] = tok; p.beam_idx[b * nb + slot] = src;
        ++slot;
      }
    }
    p.hyp_n[b] = hn; p.hyp_worst[b] = worst;
    n_cp = ncp;
    // BeamHypotheses.is_done with the best candidate of this step
    bool fin = false;
    if (hn >= nb) fin = p.early_stopping ? true : worst >= (double)cand_s[0] / denom;
    if (fin) p.done[b] = 1;
  }
  __syncthreads();
  for (int c = 0; c < n_cp; ++c) {
    const int* src = p.seq + (size_t)cp_src[c] * p.seq_ld;
    int* dst = p.hyp_seq + ((size_t)b * (BEAM_MAX + 1) + cp_slot[c]) * p.seq_ld;
    for (int t = tid; t < n; t += 1024) dst[t] = src[t];
  }
}

int beam_select_forward(const BeamState& s, hipStream_t st) {
  IDX_CHECK(s.proc && s.next_tok && s.beam_idx && s.seq && s.hyp_score && s.hyp_len && s.hyp_slot && s.hyp_seq && s.hyp_n && s.hyp_worst && s.st, "null pointer");
  hipLaunchKernelGGL(beam_select_kernel, dim3(s.B), dim3(1024), 0, st, s);
  IDX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void beam_reorder_rows_kernel(const BeamState p) {
  const int b = blockIdx.x, tid = threadIdx.x, nb = p.nb, V = p.V;
  const int n = p.st->step;
  if (p.done[b]) {
    if (tid < nb) { p.cur_tok[b * nb + tid] = p.stop_token; if (n < p.seq_ld) p.seq[(size_t)(b * nb + tid) * p.seq_ld + n] = p.stop_token; }
    return;
  }
  int src[BEAM_MAX];
  bool identity = true;
  for (int j = 0; j < nb; ++j) { src[j] = p.beam_idx[b * nb + j]; identity = identity && src[j] == b * nb + j; }
  if (!identity) {      // a thread owns one column of all nb rows: reads every source before it writes (in place)
    for (int t = tid; t < n; t += 1024) {
      int v[BEAM_MAX];
      for (int j = 0; j < nb; ++j) v[j] = p.seq[(size_t)src[j] * p.seq_ld + t];
      for (int j = 0; j < nb; ++j) p.seq[(size_t)(b * nb + j) * p.seq_ld + t] = v[j];
    }
    for (int t = tid; t < V; t += 1024) {
      unsigned char v[BEAM_MAX];
      for (int j = 0; j < nb; ++j) v[j] = p.seen[(size_t)src[j] * V + t];
      for (int j = 0; j < nb; ++j) p.seen[(size_t)(b * nb + j) * V + t] = v[j];
    }
  }
  __syncthreads();
  if (tid < nb) {
    const int r = b * nb + tid, tok = p.next_tok[r];
    p.seq[(size_t)r * p.seq_ld + n] = tok;
    p.seen[(size_t)r * V + tok] = 1;
    p.cur_tok[r] = tok;
  }
}

// KV rows of the generated positions [prompt_len, pos]: 16-byte granules (G per key: 16 in a fp32 cache, 8 in a bf16 one), a thread moves
// one granule of all nb beams
__global__ __launch_bounds__(256) void beam_reorder_kv_kernel(const BeamState p) {
  const int b = blockIdx.y, nb = p.nb;
  if (p.done[b]) return;
  int src[BEAM_MAX];
  bool identity = true;
  for (int j = 0; j < nb; ++j) { src[j] = p.beam_idx[b * nb + j]; identity = identity && src[j] == b * nb + j; }
  if (identity) return;
  const int pos = p.st->pos;                       // position written by this step's attention
  const int npos = pos - p.prompt_len + 1;
  if (npos <= 0) return;
  const int R = p.B * nb, H = p.H, Smax = p.Smax, G = p.kv_gran;
  f32x4* const kc = static_cast<f32x4*>(p.kcache);
  f32x4* const vc = static_cast<f32x4*>(p.vcache);
  const long items = (long)p.L * H * G * npos;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int lh = (int)(it / (G * npos)), rem = (int)(it - (long)lh * G * npos);
    const int l = lh / H, h = lh - l * H;
    {   // K: [L][R][H][G][Smax] granules
      const int c = rem / npos, s = p.prompt_len + (rem - c * npos);
      f32x4 v[BEAM_MAX];
      for (int j = 0; j < nb; ++j) v[j] = kc[((((size_t)l * R + src[j]) * H + h) * G + c) * Smax + s];
      for (int j = 0; j < nb; ++j) kc[((((size_t)l * R + b * nb + j) * H + h) * G + c) * Smax + s] = v[j];
    }
    {   // V: [L][R][H][Smax][G] granules
      const int s = p.prompt_len + rem / G, k = rem - (rem / G) * G;
      f32x4 v[BEAM_MAX];
      for (int j = 0; j < nb; ++j) v[j] = vc[((((size_t)l * R + src[j]) * H + h) * Smax + s) * G + k];
      for (int j = 0; j < nb; ++j) vc[((((size_t)l * R + b * nb + j) * H + h) * Smax + s) * G + k] = v[j];
    }
  }
}

int beam_reorder_forward(const BeamState& s, hipStream_t st) {
  IDX_CHECK(s.seq && s.seen && s.cur_tok && s.kcache && s.vcache && s.beam_idx && s.next_tok, "null pointer");
  hipLaunchKernelGGL(beam_reorder_rows_kernel, dim3(s.B), dim3(1024), 0, st, s);
  IDX_LAUNCH_CHECK();
  static const int cat = prof_register("beam_select_kernel + beam_reorder_rows_kernel + beam_reorder_kv_kernel");
  ProfScope prof(cat, st, 0.0, 0.0);
  hipLaunchKernelGGL(beam_reorder_kv_kernel, dim3(256, s.B), dim3(256), 0, st, s);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
