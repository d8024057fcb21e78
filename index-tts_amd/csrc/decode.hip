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
// or, with DecodeAttnArgs::kv16 (decode_attn16_kernel), the same in bf16: K [B][H][8][Smax][8], V [B][H][Smax][64].
#include <algorithm>
#include <cstdlib>

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

// Output of one (utterance, head) workgroup: o = this thread's unnormalised output element (threads 0..63), mx / l = the piece's
// score maximum and exp-sum.  NS == 1: normalise and write the A-fragment image for the c_proj GEMV.  Key split: leave (o, max, sum)
// of this piece; the last piece to arrive merges all of them in piece order.
__device__ __forceinline__ void decode_attn_finish(const DecodeAttnArgs& p, const int b, const int h, const int z, const int NS, const int tid,
                                                   const float o, const float mx, const float l) {
  const int d = p.d;
  auto put = [&](float val) {      // input of the c_proj GEMV: an A-fragment image (gemv_fx.hip) or a plain fp32 row (gemv_pl.hip)
    if (p.out_row) p.out_row[(size_t)b * d + h * 64 + tid] = val;
    else p.out[frag_index(b, h * 64 + tid, d >> 4)] = val;
  };
  if (NS == 1) {
    if (tid < 64) put(l > 0.f ? o / l : 0.f);
    return;
  }
  // ---- key split: leave (o, max, sum) of this piece; the last piece to arrive merges all of them in piece order ----
  __shared__ int s_last;
  float* mine = p.part + ((size_t)(b * p.H + h) * NS + z) * 66;
  if (tid < 64) __hip_atomic_store(&mine[tid], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 64) __hip_atomic_store(&mine[64], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 65) __hip_atomic_store(&mine[65], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // device-scope stores acknowledged before the arrival (see gemv_fx.hip)
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(&p.cnt[b * p.H + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == (unsigned)NS - 1u;
    if (s_last) __hip_atomic_store(&p.cnt[b * p.H + h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last || tid >= 64) return;
  const float* all = p.part + (size_t)(b * p.H + h) * NS * 66;
  float mi[16], li[16], oi[16];       // every piece's (max, sum, this lane's output) in ONE round trip
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const bool on = i < NS;
    mi[i] = on ? __hip_atomic_load(&all[i * 66 + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1e30f;
    li[i] = on ? __hip_atomic_load(&all[i * 66 + 65], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    oi[i] = on ? __hip_atomic_load(&all[i * 66 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  }
  float M = -1e30f;
#pragma unroll
  for (int i = 0; i < 16; ++i) M = fmaxf(M, mi[i]);
  float O = 0.f, L = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float wgt = li[i] > 0.f ? expf(mi[i] - M) : 0.f;
    O += wgt * oi[i];
    L += wgt * li[i];
  }
  put(L > 0.f ? O / L : 0.f);
}

// NT threads per workgroup: NT / 16 key groups in the P.V phase, NT keys per pass of the score phase.  512 threads at <= 128
// VGPRs (two workgroups per CU) halve the number of dependent load -> use passes of the 256-thread form.
template <int NT>
__global__ __launch_bounds__(NT, (NT == 512 ? 4 : 1)) void decode_attn_kernel(const DecodeAttnArgs p) {
  constexpr int NW = NT / 64, NG = NT / 16;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm;            // [64] scaled query
  float* knew = sm + 64;     // [64]
  float* vnew = sm + 128;    // [64]
  float* red = sm + 192;     // [2 * NW]
  float* outp = sm + 256;    // [NG][64] per-key-group partial outputs
  float* pr = sm + 256 + NG * 64;   // [Smax] scores / probabilities

  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pos = p.st->pos;                  // index of the token being processed = keys already cached
  const int d = p.d, Smax = p.Smax;
  float* kc = static_cast<float*>(p.kcache) + (size_t)(b * p.H + h) * 16 * Smax * 4;
  float* vc = static_cast<float*>(p.vcache) + (size_t)(b * p.H + h) * Smax * 64;

  // Everything that does not depend on the new token's q is put in flight first: this thread's first key (16 x 16 B,
  // coalesced across threads) and its first 8 value rows of the P.V phase; the kernel is a chain of dependent
  // HBM / L2 round trips, so the cache streams have to overlap the qkv fetch and the softmax barriers.
  // key range of this workgroup: all of [kstart, pos] (pos = the new token), or one of gridDim.z contiguous pieces of it
  const int ks0 = p.kstart ? p.kstart[b] : 0;
  const int NS = gridDim.z, z = blockIdx.z;
  const int chunk = NS > 1 ? (((pos - ks0 + NS) / NS + 15) & ~15) : pos - ks0 + 1;
  const int ks = ks0 + z * chunk;
  const int ke = min(pos, ks + chunk - 1);          // inclusive; ks > ke: an empty piece
  const int grp = tid >> 4, l16 = tid & 15;
  const int s_first = ks + tid;
  f32x4 kk0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    kk0[i] = (s_first < pos && s_first <= ke) ? *reinterpret_cast<const f32x4*>(kc + ((size_t)i * Smax + s_first) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int VP = NT == 512 ? 4 : 8;      // value rows prefetched per lane before the scores (NG * VP = 128 keys either way)
  f32x4 vpre[VP];
#pragma unroll
  for (int j = 0; j < VP; ++j) {
    const int sj = ks + grp + NG * j;
    vpre[j] = (sj < pos && sj <= ke) ? *reinterpret_cast<const f32x4*>(vc + (size_t)sj * 64 + 4 * l16) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    else if (which == 1) { knew[dd] = acc; if (z == 0) kc[((size_t)(dd >> 2) * Smax + pos) * 4 + (dd & 3)] = acc; }
    else { vnew[dd] = acc; if (z == 0) vc[(size_t)pos * 64 + dd] = acc; }
  }
  __syncthreads();

  // the query is wave-uniform: 64 SGPRs (v_readlane of one LDS read per lane) instead of 64 VGPRs
  const float qlane = qs[lane];
  float qv[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) qv[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qlane), i));

  // ---- scores: one key per thread and pass; the first pass consumes the prefetched key ----
  float mx = -1e30f;
  if (s_first <= ke) {
    float dot = 0.f;
    if (s_first == pos) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += qv[4 * i] * knew[4 * i] + qv[4 * i + 1] * knew[4 * i + 1] + qv[4 * i + 2] * knew[4 * i + 2] + qv[4 * i + 3] * knew[4 * i + 3];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += qv[4 * i] * kk0[i][0] + qv[4 * i + 1] * kk0[i][1] + qv[4 * i + 2] * kk0[i][2] + qv[4 * i + 3] * kk0[i][3];
    }
    pr[s_first] = dot;
    mx = dot;
  }
  for (int s = s_first + NT; s <= ke; s += NT) {
    float dot = 0.f;
    if (s == pos) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += qv[4 * i] * knew[4 * i] + qv[4 * i + 1] * knew[4 * i + 1] + qv[4 * i + 2] * knew[4 * i + 2] + qv[4 * i + 3] * knew[4 * i + 3];
    } else {
      f32x4 kk[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) kk[i] = *reinterpret_cast<const f32x4*>(kc + ((size_t)i * Smax + s) * 4);
#pragma unroll
      for (int i = 0; i < 16; ++i) dot += qv[4 * i] * kk[i][0] + qv[4 * i + 1] * kk[i][1] + qv[4 * i + 2] * kk[i][2] + qv[4 * i + 3] * kk[i][3];
    }
    pr[s] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
  for (int s = s_first; s <= ke; s += NT) {
    const float e = expf(pr[s] - mx);
    pr[s] = e;
    sum += e;
  }
  sum = wave_add(sum);
  if (lane == 0) red[NW + wave] = sum;
  __syncthreads();
  float l = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) l += red[NW + w];

  // ---- P.V : 16 key groups x 16 lanes; a lane owns 4 head dims (one 16-byte load per key, 256-byte rows coalesced),
  //      8 keys in flight per lane; group g takes keys ks+g, ks+g+16, ... ----
  const f32x4 vn4 = *reinterpret_cast<const f32x4*>(&vnew[4 * l16]);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
  for (int j = 0; j < VP; ++j) {
    const int sj = ks + grp + NG * j;
    if (sj <= ke) {
      const f32x4 t = pr[sj] * (sj == pos ? vn4 : vpre[j]);
      if ((j & 3) == 0) a0 += t; else if ((j & 3) == 1) a1 += t; else if ((j & 3) == 2) a2 += t; else a3 += t;
    }
  }
  for (int sb = ks + grp + VP * NG; sb <= ke; sb += 8 * NG) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + NG * j;
      v[j] = sj < pos ? *reinterpret_cast<const f32x4*>(vc + (size_t)sj * 64 + 4 * l16) : vn4;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + NG * j;
      if (sj <= ke) {
        const f32x4 t = pr[sj] * v[j];
        if ((j & 3) == 0) a0 += t; else if ((j & 3) == 1) a1 += t; else if ((j & 3) == 2) a2 += t; else a3 += t;
      }
    }
  }
  *reinterpret_cast<f32x4*>(&outp[grp * 64 + 4 * l16]) = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float o = 0.f;
  if (tid < 64) {
#pragma unroll
    for (int g = 0; g < NG; ++g) o += outp[g * 64 + tid];
  }
  decode_attn_finish(p, b, h, z, NS, tid, o, mx, l);
}

// ---- bf16 cache form (DecodeAttnArgs::kv16) ----
// Same structure, half the bytes: K [B][H][8][Smax][8] bf16 (a key's 16-byte granule per 8-dim chunk: 8 coalesced 16-byte loads per
// key instead of 16), V [B][H][Smax][64] bf16 (128-byte rows: 8 lanes x 16 bytes per key, 64 key groups per workgroup).  A key /
// value is rounded to bf16 (nearest even) when it is produced -- the new token's own k / v too, so a position reads the same
// whether it is the newest or an old one -- and widened exactly (<< 16) when used; every product and sum is fp32.
__device__ __forceinline__ unsigned bf16_rne_bits(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_round_f32(float f) { return __uint_as_float(bf16_rne_bits(f) << 16); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(NT, (NT == 512 ? 4 : 1)) void decode_attn16_kernel(const DecodeAttnArgs p) {
  constexpr int NW = NT / 64, NG = NT / 8;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm;            // [64] scaled query
  float* knew = sm + 64;     // [64] (rounded)
  float* vnew = sm + 128;    // [64] (rounded)
  float* red = sm + 192;     // [2 * NW]
  float* outp = sm + 256;    // [NG][64] per-key-group partial outputs
  float* pr = sm + 256 + NG * 64;   // [Smax] scores / probabilities

  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pos = p.st->pos;
  const int d = p.d, Smax = p.Smax;
  u32x4* const kc = static_cast<u32x4*>(p.kcache) + (size_t)(b * p.H + h) * 8 * Smax;      // granule (c, s) at c * Smax + s
  u32x4* const vc = static_cast<u32x4*>(p.vcache) + (size_t)(b * p.H + h) * Smax * 8;      // granule (s, c) at s * 8 + c

  const int ks0 = p.kstart ? p.kstart[b] : 0;
  const int NS = gridDim.z, z = blockIdx.z;
  const int chunk = NS > 1 ? (((pos - ks0 + NS) / NS + 15) & ~15) : pos - ks0 + 1;
  const int ks = ks0 + z * chunk;
  const int ke = min(pos, ks + chunk - 1);          // inclusive; ks > ke: an empty piece
  const int grp = tid >> 3, l8 = tid & 7;
  const int s_first = ks + tid;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 kk0[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kk0[i] = (s_first < pos && s_first <= ke) ? kc[(size_t)i * Smax + s_first] : zero4;
  constexpr int VP = 2;      // value rows prefetched per lane before the scores (NG * VP = 128 keys, as in the fp32 form)
  u32x4 vpre[VP];
#pragma unroll
  for (int j = 0; j < VP; ++j) {
    const int sj = ks + grp + NG * j;
    vpre[j] = (sj < pos && sj <= ke) ? vc[(size_t)sj * 8 + l8] : zero4;
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
    else {
      const unsigned bits = bf16_rne_bits(acc);
      unsigned short* dst = which == 1 ? reinterpret_cast<unsigned short*>(kc + (size_t)(dd >> 3) * Smax + pos) + (dd & 7)
                                       : reinterpret_cast<unsigned short*>(vc + (size_t)pos * 8) + dd;
      (which == 1 ? knew : vnew)[dd] = __uint_as_float(bits << 16);
      if (z == 0) *dst = (unsigned short)bits;
    }
  }
  __syncthreads();

  const float qlane = qs[lane];
  float qv[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) qv[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qlane), i));

  auto dot_new = [&]() {
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) dot += qv[i] * knew[i];
    return dot;
  };
  auto dot_key = [&](const u32x4 (&kk)[8]) {
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dot += qv[8 * i + 2 * j] * __uint_as_float(kk[i][j] << 16) + qv[8 * i + 2 * j + 1] * __uint_as_float(kk[i][j] & 0xffff0000u);
    return dot;
  };
  // ---- scores: one key per thread and pass; the first pass consumes the prefetched key ----
  float mx = -1e30f;
  if (s_first <= ke) {
    const float dot = s_first == pos ? dot_new() : dot_key(kk0);
    pr[s_first] = dot;
    mx = dot;
  }
  for (int s = s_first + NT; s <= ke; s += NT) {
    float dot;
    if (s == pos) dot = dot_new();
    else {
      u32x4 kk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kk[i] = kc[(size_t)i * Smax + s];
      dot = dot_key(kk);
    }
    pr[s] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
  for (int s = s_first; s <= ke; s += NT) {
    const float e = expf(pr[s] - mx);
    pr[s] = e;
    sum += e;
  }
  sum = wave_add(sum);
  if (lane == 0) red[NW + wave] = sum;
  __syncthreads();
  float l = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) l += red[NW + w];

  // ---- P.V : NG key groups x 8 lanes; a lane owns 8 head dims (one 16-byte load per key), 8 keys in flight per lane ----
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
  auto fma8 = [&](float (&acc)[8], const float pw, const u32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += pw * __uint_as_float(v[j] << 16);
      acc[2 * j + 1] += pw * __uint_as_float(v[j] & 0xffff0000u);
    }
  };
  auto fma8_new = [&](float (&acc)[8], const float pw) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += pw * vnew[8 * l8 + e];
  };
#pragma unroll
  for (int j = 0; j < VP; ++j) {
    const int sj = ks + grp + NG * j;
    if (sj <= ke) {
      if (sj == pos) fma8_new((j & 1) ? a1 : a0, pr[sj]);
      else fma8((j & 1) ? a1 : a0, pr[sj], vpre[j]);
    }
  }
  for (int sb = ks + grp + VP * NG; sb <= ke; sb += 8 * NG) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + NG * j;
      v[j] = (sj < pos && sj <= ke) ? vc[(size_t)sj * 8 + l8] : zero4;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = sb + NG * j;
      if (sj <= ke) {
        if (sj == pos) fma8_new((j & 1) ? a1 : a0, pr[sj]);
        else fma8((j & 1) ? a1 : a0, pr[sj], v[j]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) outp[grp * 64 + 8 * l8 + e] = a0[e] + a1[e];
  __syncthreads();
  float o = 0.f;
  if (tid < 64) {
#pragma unroll 16
    for (int g = 0; g < NG; ++g) o += outp[g * 64 + tid];
  }
  decode_attn_finish(p, b, h, z, NS, tid, o, mx, l);
}

int decode_attn_nsplit(int B, int H) {
  const int wgs = B * H;
  // (bf16 cache, 320 workgroups: two / three key pieces per (utterance, head) measured 185 / 215 ms per step against 156 unsplit)
  return wgs <= 128 ? std::max(1, std::min(16, 256 / wgs)) : 1;
}

int decode_attn_forward(const DecodeAttnArgs& a, hipStream_t stream) {
  IDX_CHECK(a.qkv_part && a.kcache && a.vcache && (a.out || a.out_row) && a.st, "null pointer");
  IDX_CHECK(a.d == a.H * 64, "head_dim must be 64");
  constexpr int nt = 512;      // 512 threads measured 4 % faster than 256 (profiles/README.md)
  const size_t lds = (size_t)(256 + (nt / (a.kv16 ? 8 : 16)) * 64 + a.Smax) * sizeof(float);
  IDX_CHECK(lds <= 128 * 1024, "Smax too large for the LDS score buffer");
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn16_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_set = true;
  }
  IDX_CHECK(a.nsplit >= 1 && a.nsplit <= 16 && (a.nsplit == 1 || (a.part && a.cnt)), "key split: 1..16 pieces, partial buffer and counters");
  // algorithmic bytes: K and V of every cached position, all heads: B * S * 2 * d * (4 | 2) (S as the host knows it: pos_hint)
  static const int cat = prof_register("decode_attn_kernel"), cat16 = prof_register("decode_attn16_kernel");
  ProfScope prof(a.kv16 ? cat16 : cat, stream, 0.0, (a.kv16 ? 4.0 : 8.0) * a.B * (double)a.pos_hint * a.d);
  if (a.kv16) hipLaunchKernelGGL(decode_attn16_kernel<512>, dim3(a.H, a.B, a.nsplit), dim3(512), lds, stream, a);
  else hipLaunchKernelGGL(decode_attn_kernel<512>, dim3(a.H, a.B, a.nsplit), dim3(512), lds, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// x[b] = mel_emb[tok] + mel_pos[mp] for the plane-GEMV decode step: fp32 row + (mean, M2) per 16 columns
template <int NT>
__device__ __forceinline__ void embed_row_pl(float* x_row, float* x_stats, const int b, const int B, const int d, const float* mel_emb,
                                             const float* mel_pos, const int tok, const int mp, const int tid) {
  const int R = ((B + 15) >> 4) * 16;
  for (int e0 = 0; e0 < d; e0 += NT) {      // d % 16 == 0: a 16-column tile never straddles the loop's edge or a wave
    const int e = e0 + tid;
    const bool on = e < d;
    const float v = on ? mel_emb[(size_t)tok * d + e] + mel_pos[(size_t)mp * d + e] : 0.f;
    float s = v;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s * 0.0625f;
    float q = (v - mean) * (v - mean);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
    if (on) {
      x_row[(size_t)b * d + e] = v;
      if ((e & 15) == 0) *reinterpret_cast<float2*>(&x_stats[((size_t)(e >> 4) * R + b) * 2]) = make_float2(mean, q);
    }
  }
}

__global__ __launch_bounds__(1024) void sample_greedy_kernel(const SampleArgs p) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  __shared__ int s_tok;
  const bool fused = p.embed.x_row || p.embed.x_frag;
  const int mp_next = fused ? p.st->mel_pos + 1 : 0;      // read before anybody can advance the state
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
    const int step = p.st->step;
    p.codes[(size_t)b * p.codes_ld + step] = tok;
    if (p.forced) tok = (int)p.forced[(size_t)b * p.forced_ld + step];      // teacher forcing: the given token continues the sequence
    p.seen[(size_t)b * V + tok] = 1;
    if (tok == p.stop_token) p.finished[b] = 1;
    p.cur_tok[b] = tok;
    s_tok = tok;
  }
  if (!fused) return;
  __syncthreads();
  if (p.embed.x_frag) {      // fp32-MFMA GEMV step: x as A-fragment images (the folded LayerNorm's statistics are computed by the consumer)
    const int d = p.embed.d, tok = s_tok;
    for (int e = tid; e < d; e += 1024) p.embed.x_frag[frag_index(b, e, d >> 4)] = p.embed.mel_emb[(size_t)tok * d + e] + p.embed.mel_pos[(size_t)mp_next * d + e];
  } else {
    embed_row_pl<1024>(p.embed.x_row, p.embed.x_stats, b, p.B, p.embed.d, p.embed.mel_emb, p.embed.mel_pos, s_tok, mp_next, tid);
  }
  __syncthreads();      // every read of the step scalars by this workgroup is behind us
  if (tid == 0) {
    DecodeState* st = p.embed.st_rw;
    const unsigned old = __hip_atomic_fetch_add(&st->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1u) {      // everybody has read pos / mel_pos / step: advance them for the next launch
      __hip_atomic_store(&st->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st->pos += 1; st->mel_pos += 1; st->step += 1;
    }
  }
}

int sample_greedy_forward(const SampleArgs& a, hipStream_t stream) {
  IDX_CHECK(a.part && a.seen && a.finished && a.codes && a.cur_tok && a.st, "null pointer");
  static const int cat = prof_register("sample_greedy_kernel");
  ProfScope prof(cat, stream, 0.0, 4.0 * a.B * (double)a.V * (a.parts + 1));
  hipLaunchKernelGGL(sample_greedy_kernel, dim3(a.B), dim3(1024), 0, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// -------------------------------------------------------------------------------------------------
// multinomial sampling with the HF warpers / the accel-engine sampler (SampleWarpArgs, decode.h)
constexpr int SW_NPT = 16;          // vocabulary entries per thread: V <= 16384
constexpr int SW_CAP = 2048;        // survivors of the top-k filter handled by the top-p stage (more only on massive ties)

__device__ __forceinline__ void block_argmax(float& v, int& i, float* rv, int* ri, int tid) {
  // max value, smallest index on ties; result broadcast to every thread
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

__global__ __launch_bounds__(1024) void sample_warp_kernel(const SampleWarpArgs q) {
  const SampleArgs& p = q.base;
  __shared__ float rv[16];
  __shared__ int ri[16];
  __shared__ float sval[SW_CAP];
  __shared__ int sidx[SW_CAP];
  __shared__ float sorted_v[SW_CAP];
  __shared__ int sorted_i[SW_CAP];
  __shared__ int s_count, s_keep_from;
  __shared__ float s_sum;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = p.V;
  const unsigned char* seen = p.seen + (size_t)b * V;
  const float* prow = p.part + (size_t)b * V;
  const size_t nbase = ((size_t)p.st->step * p.B + b) * V;
  auto draw = [&](int v) { return exp1_draw(q.exp_noise, q.seed, nbase + v); };

  // ---- scores: logits -> (repetition penalty) -> / temperature ----
  float sc[SW_NPT];
#pragma unroll
  for (int u = 0; u < SW_NPT; ++u) {
    const int v = tid + 1024 * u;
    float l = -INFINITY;
    if (v < V) {
      l = prow[v] + (p.bias ? p.bias[v] : 0.0f);
      if (p.logits_out) p.logits_out[(size_t)b * V + v] = l;
      if (q.mode == SAMPLE_HF && p.penalty != 1.0f && seen[v]) l = l < 0.f ? l * p.penalty : l / p.penalty;
      if (q.temperature != 1.0f) l = l / q.temperature;
    }
    sc[u] = l;
  }

  int token;
  if (q.mode == SAMPLE_ACCEL || (q.top_k == 0 && q.top_p >= 1.0f)) {
    // softmax over the whole row, divided by the (accel sampler: clamped) noise, argmax
    const float qmin = q.mode == SAMPLE_ACCEL ? 1e-10f : 0.0f;
    float mx = -INFINITY; int mi = 0;
#pragma unroll
    for (int u = 0; u < SW_NPT; ++u) if (sc[u] > mx) { mx = sc[u]; mi = tid + 1024 * u; }
    block_argmax(mx, mi, rv, ri, tid);
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < SW_NPT; ++u) if (tid + 1024 * u < V) part += expf(sc[u] - mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    __syncthreads();
    if ((tid & 63) == 0) rv[tid >> 6] = part;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += rv[w];
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < SW_NPT; ++u) {
      const int v = tid + 1024 * u;
      if (v < V) {
        const float r = (expf(sc[u] - mx) / tot) / fmaxf(draw(v), qmin);
        if (r > best) { best = r; bi = v; }
      }
    }
    block_argmax(best, bi, rv, ri, tid);
    token = bi;
  } else {
    // ---- top-k: the k-th largest value (with multiplicity) by k rounds of block-wide max extraction ----
    if (q.top_k > 0 && q.top_k < V) {
      unsigned taken = 0;
      float kth = -INFINITY;
      for (int r = 0; r < q.top_k; ++r) {
        float mx = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < SW_NPT; ++u) {
          const int v = tid + 1024 * u;
          if (v < V && !((taken >> u) & 1u) && (sc[u] > mx || (sc[u] == mx && v < mi))) { mx = sc[u]; mi = v; }
        }
        block_argmax(mx, mi, rv, ri, tid);
        kth = mx;
        if ((mi & 1023) == tid && mi < V) taken |= 1u << (mi >> 10);
      }
#pragma unroll
      for (int u = 0; u < SW_NPT; ++u) if (sc[u] < kth) sc[u] = -INFINITY;
    }
    // ---- survivors -> LDS (finite scores only: -inf has probability 0 and sorts first) ----
    if (tid == 0) s_count = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SW_NPT; ++u) {
      const int v = tid + 1024 * u;
      if (v < V && sc[u] > -INFINITY) {
        const int slot = atomicAdd(&s_count, 1);
        if (slot < SW_CAP) { sval[slot] = sc[u]; sidx[slot] = v; }
      }
    }
    __syncthreads();
    const int n = min(s_count, SW_CAP);
    // rank sort, ascending by (value, index)
    for (int e = tid; e < n; e += 1024) {
      const float ve = sval[e]; const int ie = sidx[e];
      int rank = 0;
      for (int o = 0; o < n; ++o) rank += (sval[o] < ve || (sval[o] == ve && sidx[o] < ie)) ? 1 : 0;
      sorted_v[rank] = ve; sorted_i[rank] = ie;
    }
    __syncthreads();
    // ---- top-p on the ascending list: softmax, running sum, remove while cum <= 1 - top_p (never the last one) ----
    if (tid == 0) {
      int keep_from = 0;
      const float mx = sorted_v[n - 1];
      if (q.top_p < 1.0f) {
        float tot = 0.f;
        for (int e = 0; e < n; ++e) tot += expf(sorted_v[e] - mx);
        const float thr = (float)(1.0 - (double)q.top_p);
        float cum = 0.f;
        for (int e = 0; e < n - 1; ++e) {
          cum += expf(sorted_v[e] - mx) / tot;
          if (cum <= thr) keep_from = e + 1; else break;
        }
      }
      float tot2 = 0.f;      // softmax denominator of the warped scores
      for (int e = keep_from; e < n; ++e) tot2 += expf(sorted_v[e] - mx);
      s_keep_from = keep_from;
      s_sum = tot2;
    }
    __syncthreads();
    // ---- multinomial == argmax(probs / q) over the kept tokens ----
    const float mx = sorted_v[n - 1];
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int e = s_keep_from + tid; e < n; e += 1024) {
      const int v = sorted_i[e];
      const float r = (expf(sorted_v[e] - mx) / s_sum) / draw(v);
      if (r > best || (r == best && v < bi)) { best = r; bi = v; }
    }
    block_argmax(best, bi, rv, ri, tid);
    token = bi;
  }
  if (tid == 0) {
    const int tok = p.finished[b] ? p.stop_token : token;
    p.codes[(size_t)b * p.codes_ld + p.st->step] = tok;
    p.seen[(size_t)b * V + tok] = 1;
    if (tok == p.stop_token) p.finished[b] = 1;
    p.cur_tok[b] = tok;
  }
}

int sample_warp_forward(const SampleWarpArgs& a, hipStream_t stream) {
  const SampleArgs& b = a.base;
  IDX_CHECK(b.part && b.parts == 1 && b.seen && b.finished && b.codes && b.cur_tok && b.st, "null pointer");
  IDX_CHECK(b.V > 0 && b.V <= 1024 * SW_NPT, "vocabulary size");
  IDX_CHECK(a.temperature > 0.0f && a.top_k >= 0 && a.top_p > 0.0f, "sampling parameters");
  IDX_CHECK(a.mode == SAMPLE_HF || a.mode == SAMPLE_ACCEL, "sampling mode");
  static const int cat = prof_register("sample_warp_kernel");
  ProfScope prof(cat, stream, 0.0, 8.0 * b.B * (double)b.V);
  hipLaunchKernelGGL(sample_warp_kernel, dim3(b.B), dim3(1024), 0, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

__global__ void advance_state_kernel(DecodeState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->pos += 1; st->mel_pos += 1; st->step += 1; }
}

int advance_state(DecodeState* st, hipStream_t stream) {
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(64), 0, stream, st);
  IDX_LAUNCH_CHECK();
  return 0;
}

// qkv [B][S][3d] (token-major, after bias) -> K/V caches, positions [0, S)
template <bool KV16>
__global__ __launch_bounds__(256) void kv_store_prefill_kernel(float* qkv, void* kcache, void* vcache, int B, int H, int S, int Smax, int d) {
  const int s = blockIdx.x, b = blockIdx.y;
  float* row = qkv + ((size_t)b * S + s) * 3 * d;
  for (int col = threadIdx.x; col < d; col += 256) {
    const int h = col >> 6, dd = col & 63;
    if (KV16) {      // bf16 cache; the fp32 row keeps the rounded values for the prefill attention
      const unsigned kb = bf16_rne_bits(row[d + col]), vb = bf16_rne_bits(row[2 * d + col]);
      static_cast<unsigned short*>(kcache)[(((size_t)(b * H + h) * 8 + (dd >> 3)) * Smax + s) * 8 + (dd & 7)] = (unsigned short)kb;
      static_cast<unsigned short*>(vcache)[((size_t)(b * H + h) * Smax + s) * 64 + dd] = (unsigned short)vb;
      row[d + col] = __uint_as_float(kb << 16);
      row[2 * d + col] = __uint_as_float(vb << 16);
    } else {
      static_cast<float*>(kcache)[(((size_t)(b * H + h) * 16 + (dd >> 2)) * Smax + s) * 4 + (dd & 3)] = row[d + col];
      static_cast<float*>(vcache)[((size_t)(b * H + h) * Smax + s) * 64 + dd] = row[2 * d + col];
    }
  }
}

int kv_store_prefill(float* qkv, void* kcache, void* vcache, int kv16, int B, int H, int S, int Smax, int d, hipStream_t stream) {
  IDX_CHECK(S <= Smax, "prefill longer than the cache");
  static const int cat = prof_register("kv_store_prefill_kernel");
  ProfScope prof(cat, stream, 0.0, (kv16 ? 28.0 : 16.0) * B * (double)S * d);
  if (kv16) hipLaunchKernelGGL(kv_store_prefill_kernel<true>, dim3(S, B), dim3(256), 0, stream, qkv, kcache, vcache, B, H, S, Smax, d);
  else hipLaunchKernelGGL(kv_store_prefill_kernel<false>, dim3(S, B), dim3(256), 0, stream, qkv, kcache, vcache, B, H, S, Smax, d);
  IDX_LAUNCH_CHECK();
  return 0;
}

// out[r][:] = sum over the tables t with idx[t][r] >= 0 of table[t][idx[t][r]][:]
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const GatherArgs p) {
  const int r = blockIdx.x;
  int id[GATHER_MAX_TABLES];
#pragma unroll
  for (int t = 0; t < GATHER_MAX_TABLES; ++t) {
    id[t] = (p.table[t] && p.idx[t]) ? p.idx[t][r] : -1;
    if (p.table_rows[t] > 0 && id[t] >= p.table_rows[t]) {      // nn.Embedding would raise IndexError: never read past the table
      if (p.oob && threadIdx.x == 0) *p.oob = 1 + t;
      id[t] = -1;
    }
  }
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
  static const int cat = prof_register("gather_sum_rows_kernel");
  ProfScope prof(cat, stream, 0.0, 8.0 * rows * (double)a.d);
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

__global__ __launch_bounds__(256) void embed_step_pl_kernel(float* x_row, float* x_stats, int B, int d, const float* mel_emb,
                                                            const float* mel_pos, const int* cur_tok, const DecodeState* st) {
  embed_row_pl<256>(x_row, x_stats, blockIdx.x, B, d, mel_emb, mel_pos, cur_tok[blockIdx.x], st->mel_pos, threadIdx.x);
}

int embed_step_pl(float* x_row, float* x_stats, int B, int d, const float* mel_emb, const float* mel_pos, const int* cur_tok,
                  const DecodeState* st, hipStream_t stream) {
  IDX_CHECK(d % 16 == 0, "row statistics per 16 columns need d % 16 == 0");
  static const int cat = prof_register("embed_step_kernel");
  ProfScope prof(cat, stream, 0.0, 12.0 * B * (double)d);
  hipLaunchKernelGGL(embed_step_pl_kernel, dim3(B), dim3(256), 0, stream, x_row, x_stats, B, d, mel_emb, mel_pos, cur_tok, st);
  IDX_LAUNCH_CHECK();
  return 0;
}

int embed_step(float* x, int B, int d, const float* mel_emb, const float* mel_pos, const int* cur_tok, const DecodeState* st,
               hipStream_t stream) {
  static const int cat = prof_register("embed_step_kernel");
  ProfScope prof(cat, stream, 0.0, 12.0 * B * (double)d);
  hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, stream, x, d, mel_emb, mel_pos, cur_tok, st);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
