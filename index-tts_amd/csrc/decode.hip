// Per-token kernels of the GPT autoregressive decode (the reference's `accel_engine.generate` plugin
// slot, model_v2.py:871-883 / accel/accel_engine.py:378-645, and the HF `_sample` loop it falls back to,
// transformers_generation_utils.py:3196-3269):
//   decode_attn_kernel   one query token per (utterance, head) against the KV cache (HBM-bound KV read);
//                        also folds the c_attn split-K partial sum + bias and the KV-cache write
//                        (the reference's Triton `store_kvcache_kernel`, accel/attention.py:57-86)
//   sample_greedy_kernel lm_head partial sum + bias -> RepetitionPenalty (gen_utils 900-901) -> argmax
//                        (3252) -> finished/pad bookkeeping (3255-3264), all on device: no host sync per token
//   kv_store_prefill / advance_state  glue
// All per-step scalars (cache position, mel position, output column) live in a DecodeState in HBM so a
// step is replayable (hipGraph) without re-binding kernel arguments.
//
// KV cache layout (chosen for the decode read pattern, the only hot reader):
//   K: [B][H][16][Smax][4]   dot products walk the keys with lanes = keys -> 16-byte, fully coalesced
//   V: [B][H][Smax][64]      P.V walks the keys with lanes = head dim    -> 256-byte coalesced rows
#include "decode.h"
#include "prof.h"

namespace idxtts {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm;            // [64] scaled query
  float* knew = sm + 64;     // [64]
  float* vnew = sm + 128;    // [64]
  float* red = sm + 192;     // [8]
  float* outp = sm + 256;    // [16][64] per-key-group partial outputs
  float* pr = sm + 256 + 1024;   // [Smax] scores / probabilities

  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pos = p.st->pos;                  // index of the token being processed = keys already cached
  const int d = p.d, Smax = p.Smax;
  float* kc = p.kcache + (size_t)(b * p.H + h) * 16 * Smax * 4;
  float* vc = p.vcache + (size_t)(b * p.H + h) * Smax * 64;

  // Everything that does not depend on the new token's q is put in flight first: this thread's first key (16 x 16 B,
  // coalesced across threads) and its first 8 value rows of the P.V phase; the kernel is a chain of dependent
  // HBM / L2 round trips, so the cache streams have to overlap the qkv fetch and the softmax barriers.
  const int ks = p.kstart ? p.kstart[b] : 0;
  const int grp = tid >> 4, l16 = tid & 15;
  const int s_first = ks + tid;
  f32x4 kk0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    kk0[i] = s_first < pos ? *reinterpret_cast<const f32x4*>(kc + ((size_t)i * Smax + s_first) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 vpre[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sj = ks + grp + 16 * j;
    vpre[j] = sj < pos ? *reinterpret_cast<const f32x4*>(vc + (size_t)sj * 64 + 4 * l16) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- q, k, v of the new token (waves 0,1,2 take q,k,v) ----
  if (tid < 192) {
    const int which = tid >> 6, dd = tid & 63;
    const int col = which * d + h * 64 + dd;
    const float* row = p.qkv_part + (size_t)b * 3 * d + col;
    const size_t sst = (size_t)p.part_rows * 3 * d;
    float acc = p.qkv_bias ? p.qkv_bias[col] : 0.0f;
    for (int s = 0; s < p.parts; ++s) acc += row[(size_t)s * sst];
    if (which == 0) qs[dd] = acc * p.scale;
    else if (which == 1) { knew[dd] = acc; kc[((size_t)(dd >> 2) * Smax + pos) * 4 + (dd & 3)] = acc; }
    else { vnew[dd] = acc; vc[(size_t)pos * 64 + dd] = acc; }
  }
  __syncthreads();

  f32x4 q4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) q4[i] = *reinterpret_cast<const f32x4*>(&qs[4 * i]);

  // ---- scores: one key per thread ----
  float mx = -1e30f;
  for (int s = s_first; s <= pos; s += 256) {
    float dot = 0.f;
    if (s == pos) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(&knew[4 * i]);
        dot += q4[i][0] * kk[0] + q4[i][1] * kk[1] + q4[i][2] * kk[2] + q4[i][3] * kk[3];
      }
    } else if (s == s_first) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += q4[i][0] * kk0[i][0] + q4[i][1] * kk0[i][1] + q4[i][2] * kk0[i][2] + q4[i][3] * kk0[i][3];
    } else {
      f32x4 kk[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) kk[i] = *reinterpret_cast<const f32x4*>(kc + ((size_t)i * Smax + s) * 4);
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += q4[i][0] * kk[i][0] + q4[i][1] * kk[i][1] + q4[i][2] * kk[i][2] + q4[i][3] * kk[i][3];
    }
    pr[s] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int s = s_first; s <= pos; s += 256) {
    const float e = expf(pr[s] - mx);
    pr[s] = e;
    sum += e;
  }
  sum = wave_add(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float l = red[4] + red[5] + red[6] + red[7];

  // ---- P.V : 16 key groups x 16 lanes; a lane owns 4 head dims (one 16-byte load per key, 256-byte rows coalesced),
  //      8 keys in flight per lane; group g takes keys ks+g, ks+g+16, ... ----
  const f32x4 vn4 = *reinterpret_cast<const f32x4*>(&vnew[4 * l16]);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sj = ks + grp + 16 * j;
    if (sj <= pos) {
      const f32x4 t = pr[sj] * (sj == pos ? vn4 : vpre[j]);
      if ((j & 3) == 0) a0 += t; else if ((j & 3) == 1) a1 += t; else if ((j & 3) == 2) a2 += t; else a3 += t;
    }
  }
  for (int sb = ks + grp + 128; sb <= pos; sb += 128) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + 16 * j;
      v[j] = sj < pos ? *reinterpret_cast<const f32x4*>(vc + (size_t)sj * 64 + 4 * l16) : vn4;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + 16 * j;
      if (sj <= pos) {
        const f32x4 t = pr[sj] * v[j];
        if ((j & 3) == 0) a0 += t; else if ((j & 3) == 1) a1 += t; else if ((j & 3) == 2) a2 += t; else a3 += t;
      }
    }
  }
  *reinterpret_cast<f32x4*>(&outp[grp * 64 + 4 * l16]) = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (tid < 64) {
    float o = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) o += outp[g * 64 + tid];
    p.out[frag_index(b, h * 64 + tid, d >> 4)] = l > 0.f ? o / l : 0.f;     // A-fragment image for the c_proj GEMV
  }
}

int decode_attn_forward(const DecodeAttnArgs& a, hipStream_t stream) {
  IDX_CHECK(a.qkv_part && a.kcache && a.vcache && a.out && a.st, "null pointer");
  IDX_CHECK(a.d == a.H * 64, "head_dim must be 64");
  const size_t lds = (size_t)(256 + 1024 + a.Smax) * sizeof(float);
  IDX_CHECK(lds <= 160 * 1024, "Smax too large for the LDS score buffer");
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  // algorithmic bytes depend on the device-side position; the caller (bench) accounts for them
  ProfScope prof(PROF_DECODE_ATTN, stream, 0.0, 0.0);
  hipLaunchKernelGGL(decode_attn_kernel, dim3(a.H, a.B), dim3(256), lds, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sample_greedy_kernel(const SampleArgs p) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int V = p.V;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  const unsigned char* seen = p.seen + (size_t)b * V;
  const size_t sst = (size_t)p.part_rows * V;
  const float* prow = p.part + (size_t)b * V;
  for (int v0 = tid; v0 < V; v0 += 4096) {       // 4 vocabulary entries per trip, all their slab loads in flight together
    float l4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + 1024 * u;
      l4[u] = (v < V && p.bias) ? p.bias[v] : 0.0f;
    }
    for (int s = 0; s < p.parts; ++s) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + 1024 * u;
        if (v < V) l4[u] += prow[(size_t)s * sst + v];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + 1024 * u;
      if (v >= V) continue;
      float l = l4[u];
      if (p.logits_out) p.logits_out[(size_t)b * V + v] = l;
      if (seen[v]) l = l < 0.f ? l * p.penalty : l / p.penalty;
      if (l > best) { best = l; bidx = v; }      // ascending v per thread: strict > keeps the first maximum
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (lane == 0) { rv[wave] = best; ri[wave] = bidx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (rv[w] > best || (rv[w] == best && ri[w] < bidx)) { best = rv[w]; bidx = ri[w]; }
    int tok = p.finished[b] ? p.stop_token : bidx;      // finished rows emit pad (= eos = stop token)
    p.codes[(size_t)b * p.codes_ld + p.st->step] = tok;
    p.seen[(size_t)b * V + tok] = 1;
    if (tok == p.stop_token) p.finished[b] = 1;
    p.cur_tok[b] = tok;
  }
}

int sample_greedy_forward(const SampleArgs& a, hipStream_t stream) {
  IDX_CHECK(a.part && a.seen && a.finished && a.codes && a.cur_tok && a.st, "null pointer");
  ProfScope prof(PROF_SAMPLE, stream, 0.0, 4.0 * a.B * (double)a.V * (a.parts + 1));
  hipLaunchKernelGGL(sample_greedy_kernel, dim3(a.B), dim3(1024), 0, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
__global__ void advance_state_kernel(DecodeState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->pos += 1; st->mel_pos += 1; st->step += 1; }
}

int advance_state(DecodeState* st, hipStream_t stream) {
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(64), 0, stream, st);
  IDX_LAUNCH_CHECK();
  return 0;
}

// qkv [B][S][3d] (token-major, after bias) -> K/V caches, positions [0, S)
__global__ __launch_bounds__(256) void kv_store_prefill_kernel(const float* qkv, float* kcache, float* vcache, int B, int H,
                                                               int S, int Smax, int d) {
  const int s = blockIdx.x, b = blockIdx.y;
  const float* row = qkv + ((size_t)b * S + s) * 3 * d;
  for (int col = threadIdx.x; col < d; col += 256) {
    const int h = col >> 6, dd = col & 63;
    kcache[(((size_t)(b * H + h) * 16 + (dd >> 2)) * Smax + s) * 4 + (dd & 3)] = row[d + col];
    vcache[((size_t)(b * H + h) * Smax + s) * 64 + dd] = row[2 * d + col];
  }
}

int kv_store_prefill(const float* qkv, float* kcache, float* vcache, int B, int H, int S, int Smax, int d, hipStream_t stream) {
  IDX_CHECK(S <= Smax, "prefill longer than the cache");
  ProfScope prof(PROF_ELTWISE, stream, 0.0, 16.0 * B * (double)S * d);
  hipLaunchKernelGGL(kv_store_prefill_kernel, dim3(S, B), dim3(256), 0, stream, qkv, kcache, vcache, B, H, S, Smax, d);
  IDX_LAUNCH_CHECK();
  return 0;
}

// out[r][:] = sum over the tables t with idx[t][r] >= 0 of table[t][idx[t][r]][:]
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const GatherArgs p) {
  const int r = blockIdx.x;
  int id[GATHER_MAX_TABLES];
#pragma unroll
  for (int t = 0; t < GATHER_MAX_TABLES; ++t) id[t] = (p.table[t] && p.idx[t]) ? p.idx[t][r] : -1;
  for (int e = threadIdx.x; e < p.d; e += 256) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < GATHER_MAX_TABLES; ++t)
      if (id[t] >= 0) v += p.table[t][(size_t)id[t] * p.d + e];
    p.out[(size_t)r * p.ld_out + e] = v;
  }
}

int gather_sum_rows(const GatherArgs& a, int rows, hipStream_t stream) {
  if (rows == 0) return 0;
  IDX_CHECK(a.out && a.d > 0, "gather args");
  ProfScope prof(PROF_EMBED, stream, 0.0, 8.0 * rows * (double)a.d);
  hipLaunchKernelGGL(gather_sum_rows_kernel, dim3(rows), dim3(256), 0, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// decode-step embedding: x[b] = mel_emb[cur_tok[b]] + mel_pos[st->mel_pos], written as A-fragment images (common.h)
__global__ __launch_bounds__(256) void embed_step_kernel(float* x, int d, const float* mel_emb, const float* mel_pos,
                                                         const int* cur_tok, const DecodeState* st) {
  const int b = blockIdx.x;
  const int tok = cur_tok[b], mp = st->mel_pos;
  for (int e = threadIdx.x; e < d; e += 256) x[frag_index(b, e, d >> 4)] = mel_emb[(size_t)tok * d + e] + mel_pos[(size_t)mp * d + e];
}

int embed_step(float* x, int B, int d, const float* mel_emb, const float* mel_pos, const int* cur_tok, const DecodeState* st,
               hipStream_t stream) {
  ProfScope prof(PROF_EMBED, stream, 0.0, 12.0 * B * (double)d);
  hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, stream, x, d, mel_emb, mel_pos, cur_tok, st);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
