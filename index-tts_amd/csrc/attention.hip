// Flash-style multi-head attention (head_dim 64) on the gfx950 fp32 matrix core.
//
// Replaces the reference's attention call sites on the hot path:
//   GPT-2 prefill / latent pass   causal + left-pad key mask   transformers_gpt2.py:196-234 (_attn),
//                                                               487-575 (SDPA form), mask 1052-1058
//   s2mel DiT                     non-causal + key-padding     gpt_fast/model.py:270-308 (SDPA with
//                                                               attn_mask [B,1,T,T] from x_lens, diffusion_transformer.py:235-237)
// q,k,v are read in place from a fused [B][S][*] projection buffer (token stride / head offset given),
// the S x S score matrix is never materialised.
//
// CDNA4 mapping.  One workgroup = 4 waves = 128 query rows of one (batch, head); K/V tiles of 32 keys
// are shared through LDS (register-staged double buffer, one barrier per tile).  Per wave and tile:
//   S^T = K . Q^T   (32 keys x 32 queries, 32 x v_mfma_f32_32x32x2_f32 over d = 64)
// is computed TRANSPOSED so that a query is a LANE: the online-softmax row statistics (max, sum) are
// then per-lane scalars over the 16 accumulator registers plus ONE cross-half exchange
// (__shfl_xor 32) instead of a 32-lane shuffle tree per row.  The probabilities stay in the
// accumulator registers and feed the second product directly as its B operand
//   O^T += V^T . P^T   (2 tiles of 32 d x 32 queries, k-order = accumulator row order)
// so P never touches LDS.  Q fragments live in registers for the whole kernel (pre-scaled by 1/8).
#include <algorithm>

#include "attention.h"
#include "prof.h"

namespace idxtts {

// lane <-> lane ^ 32 exchange by v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip on the softmax's critical path)
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr int KROW = 68;     // K tile row stride (floats): 64 + 4 pad -> conflict-free ds_read_b128 fragments
constexpr float NEG_BIG = -1e30f;
constexpr int REL_MAX = 96;  // rows of a relative_key table (three 32-row tiles)
constexpr int REL_ROW = 97;  // floats per query in the LDS table of q . rel_key products (odd: lanes = queries hit distinct banks)
constexpr int REL_LDS = 4 * 32 * REL_ROW * (int)sizeof(float);      // dynamic LDS of a relative_key launch: one table per wave

// scores of one 32 x 32 tile (register r of lane half h = key key0 + (r & 3) + 8 (r >> 2) + 4 h, lane j = query qi) += the relative_key
// term from the wave's table
__device__ __forceinline__ void add_rel_term(f32x16& s, const float* tbl_row, int key0, int q0, int qi, int h, int L, int R) {
  const int dmin = key0 - (q0 + 31), dmax = key0 + 31 - q0;      // range of key - query over the tile (wave-uniform)
  if (dmax <= -L || dmin >= R) {
    const float c = tbl_row[dmax <= -L ? 0 : L + R];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += c;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      s[r] += tbl_row[min(max(key - qi, -L), R) + L];
    }
  }
}

__global__ __launch_bounds__(256) void flash_attn_f32_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * KROW + 2 * 32 * 64];
  float (*Ks)[32 * KROW] = reinterpret_cast<float (*)[32 * KROW]>(smem);
  float (*Vs)[32 * 64] = reinterpret_cast<float (*)[32 * 64]>(smem + 2 * 32 * KROW);

  const int qblk = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int q0 = qblk * 128 + wave * 32;            // this wave's first query
  const int qi = q0 + j;                            // this lane's query
  const int kstart = p.kstart ? p.kstart[b] : 0;
  const int kend = p.kend ? min(p.kend[b], p.Sk) : p.Sk;
  // keys this block can see at all
  int k_hi = kend;
  if (p.causal) k_hi = min(k_hi, qblk * 128 + 128);
  const int k_lo = kstart & ~31;
  const int ntiles = k_hi > k_lo ? (k_hi - k_lo + 31) >> 5 : 0;

  const float* qb = p.q + (size_t)b * p.q_bs + hd * 64;
  const float* kb = p.k + (size_t)b * p.k_bs + hd * 64;
  const float* vb = p.v + (size_t)b * p.v_bs + hd * 64;

  // Q fragments: B operand of K.Q^T -> lane (j,h) holds Q[qi][8g + 4h + s], g = 0..7
  f32x4 qf[8];
  {
    const bool ok = qi < p.Sq;
    const float* qrow = qb + (size_t)(ok ? qi : 0) * p.q_ts;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(qrow + 8 * g + 4 * h);
      qf[g] = v * p.scale;
    }
  }

  f32x4 kr[2], vr[2];
  auto load_kv = [&](int tile) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int idx = tid + 256 * l;            // 512 float4 per 32x64 tile
      const int row = idx >> 4, c4 = idx & 15;
      const int key = k_lo + tile * 32 + row;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (key < p.Sk) {
        a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * p.k_ts + c4 * 4);
        c = *reinterpret_cast<const f32x4*>(vb + (size_t)key * p.v_ts + c4 * 4);
      }
      kr[l] = a;
      vr[l] = c;
    }
  };
  auto store_kv = [&](int buf) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<f32x4*>(&Ks[buf][row * KROW + c4 * 4]) = kr[l];
      *reinterpret_cast<f32x4*>(&Vs[buf][row * 64 + c4 * 4]) = vr[l];
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = NEG_BIG, l_run = 0.0f;

  extern __shared__ float rel_dyn[];
  float* const tbl_row = rel_dyn + (wave * 32 + j) * REL_ROW;      // this lane's query
  if (p.rel_key) {      // q . rel_key[r] for this workgroup's queries: the table's rows go through the K tile and the QK MFMAs
    const int nrel = p.rel_left + p.rel_right + 1;
    for (int rt = 0; rt * 32 < nrel; ++rt) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int idx = tid + 256 * l, row = idx >> 4, c4 = idx & 15, rr = rt * 32 + row;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (rr < nrel) a = *reinterpret_cast<const f32x4*>(p.rel_key + (size_t)rr * 64 + c4 * 4);
        *reinterpret_cast<f32x4*>(&Ks[0][row * KROW + c4 * 4]) = a;
      }
      __syncthreads();
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks[0] + j * KROW + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tbl_row[rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = s[r];
      __syncthreads();
    }
  }

  if (ntiles > 0) {
    load_kv(0);
    store_kv(0);
  }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool has_next = tile + 1 < ntiles;
    if (has_next) load_kv(tile + 1);
    const float* kt = Ks[tile & 1];
    const float* vt = Vs[tile & 1];
    const int key0 = k_lo + tile * 32;
    // wave-uniform skip: under the causal mask a tile entirely in this wave's future contributes nothing
    const bool wave_active = !(p.causal && key0 > q0 + 31) && q0 < p.Sq;
    if (wave_active) {
      // ---- S^T = K . Q^T ----
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kt + j * KROW + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], s, 0, 0, 0);
      }
      if (p.rel_key) add_rel_term(s, tbl_row, key0, q0, qi, h, p.rel_left, p.rel_right);
      // ---- mask + online softmax (query = lane, keys = registers of both lane halves) ----
      float mx = NEG_BIG;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = key >= kstart && key < kend && (!p.causal || key <= qi);
        s[r] = ok ? s[r] : NEG_BIG;
        mx = fmaxf(mx, s[r]);
      }
      mx = xor32_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = expf(m_run - m_new);       // m_run = NEG_BIG -> 0 (or 1 if still nothing seen: o,l are 0 anyway)
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = s[r] <= -1e29f ? 0.0f : expf(s[r] - m_new);
        s[r] = pv;
        psum += pv;
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      // ---- O^T += V^T . P^T : k-step r uses key (r&3)+8(r>>2)+4h, exactly the row that register r of s holds ----
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float* vrow = vt + krow * 64 + j;
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[t * 32], s[r], o[t], 0, 0, 0);
      }
    }
    if (has_next) store_kv((tile + 1) & 1);
    __syncthreads();
  }

  // ---- normalise and store: O^T tiles (rows = d, cols = query) -> O[b][q][head*64 + d] through LDS ----
  const float l_tot = xor32_sum(l_run);
  const float inv_l = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  float* stage = smem + wave * (32 * 65);            // reuse the (now idle) K/V tiles as the transpose buffer
  static_assert(4 * 32 * 65 <= 2 * 32 * KROW + 2 * 32 * 64, "output staging must fit in the K/V tiles");
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      stage[j * 65 + dd] = o[t][r] * inv_l;
    }
  __syncthreads();
  float* ob = p.o + (size_t)b * p.o_bs + hd * 64;
#pragma unroll
  for (int it = 0; it < 32; ++it) {                  // one 256-byte output row per wave-instruction
    const int qq = q0 + it;
    if (qq < p.Sq) ob[(size_t)qq * p.o_ts + lane] = stage[it * 65 + lane];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 variant for the compute-bound non-decode passes (s2mel DiT, GPT latent pass): same tiling and the same
// "query = lane" online softmax, but both products run on v_mfma_f32_32x32x16_bf16 with every fp32 operand written as
// hi + lo (bf16 each) and three MFMAs per product (hi*hi + hi*lo + lo*hi, fp32 accumulation: relative error ~2^-16):
// 24 bf16 MFMAs (768 cycles) per 32 x 32 (query, key) tile instead of 64 f32 MFMAs (4096 cycles).
//   S^T = K . Q^T : K rows (hi/lo planes, 144-byte rows) are ds_read_b128 A operands, Q^T lives in registers (pre-scaled by
//                   scale * log2 e so the softmax uses v_exp_f32 directly);
//   O^T += V^T . P^T : P stays in the accumulator registers; register r of lane half h is key (r&3) + 8(r>>2) + 4h, i.e.
//                   the 8 k-slots a lane feeds to MFMA t2 are keys 16 t2 + 4h + {0..3, 8..11}; V is kept ROW-major in LDS
//                   (coalesced 8-byte stores) and read as the matching A operand with two ds_read_b64_tr_b16 transposing
//                   reads (4 consecutive keys x 16 d each; 192-byte rows keep the four rows of a read on disjoint banks).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KB16 = 144;          // K row bytes: 64 bf16 + 8 pad
constexpr int VB16 = 192;          // V row bytes: 64 bf16 + 32 pad
constexpr int ABUF = 2 * 32 * KB16 + 2 * 32 * VB16;     // hi/lo K, hi/lo V of one 32-key tile: 21504 B

__device__ __forceinline__ void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) { split_bf16_x4(v, hi, lo); }

__global__ __launch_bounds__(256) void flash_attn_bf16x3_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ABUF];

  const int qblk = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int q0 = qblk * 128 + wave * 32;
  const int qi = q0 + j;
  const int kstart = p.kstart ? p.kstart[b] : 0;
  const int kend = p.kend ? min(p.kend[b], p.Sk) : p.Sk;
  int k_hi = kend;
  if (p.causal) k_hi = min(k_hi, qblk * 128 + 128);
  const int k_lo = kstart & ~31;
  const int ntiles = k_hi > k_lo ? (k_hi - k_lo + 31) >> 5 : 0;

  const float* qb = p.q + (size_t)b * p.q_bs + hd * 64;
  const float* kb = p.k + (size_t)b * p.k_bs + hd * 64;
  const float* vb = p.v + (size_t)b * p.v_bs + hd * 64;

  // Q^T fragments (B operand): lane (query j, half h) holds Q[qi][16c + 8h + e], e = 0..7, for the four 16-d chunks c
  bf16x8 qh[4], ql[4];
  {
    const bool ok = qi < p.Sq;
    const float* qrow = qb + (size_t)(ok ? qi : 0) * p.q_ts;
    const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, bq = a;
      if (ok) {
        a = *reinterpret_cast<const f32x4*>(qrow + 16 * c + 8 * h);
        bq = *reinterpret_cast<const f32x4*>(qrow + 16 * c + 8 * h + 4);
      }
      bf16x4 ah, al, bh, bl;
      split4(a * sc, ah, al);
      split4(bq * sc, bh, bl);
      qh[c] = __builtin_shufflevector(ah, bh, 0, 1, 2, 3, 4, 5, 6, 7);
      ql[c] = __builtin_shufflevector(al, bl, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }

  f32x4 kr[2], vr[2];
  auto load_kv = [&](int tile) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 4, c4 = idx & 15;
      const int key = k_lo + tile * 32 + row;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (key < p.Sk) {
        a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * p.k_ts + c4 * 4);
        c = *reinterpret_cast<const f32x4*>(vb + (size_t)key * p.v_ts + c4 * 4);
      }
      kr[l] = a;
      vr[l] = c;
    }
  };
  auto store_kv = [&](int buf) {
    char* base = smem + buf * ABUF;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 4, c4 = idx & 15;
      bf16x4 hi, lo;
      split4(kr[l], hi, lo);
      *reinterpret_cast<bf16x4*>(base + row * KB16 + c4 * 8) = hi;
      *reinterpret_cast<bf16x4*>(base + 32 * KB16 + row * KB16 + c4 * 8) = lo;
      split4(vr[l], hi, lo);
      *reinterpret_cast<bf16x4*>(base + 64 * KB16 + row * VB16 + c4 * 8) = hi;
      *reinterpret_cast<bf16x4*>(base + 64 * KB16 + 32 * VB16 + row * VB16 + c4 * 8) = lo;
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = NEG_BIG, l_run = 0.0f;

  // byte offsets of this lane's operand reads inside a tile buffer
  const int k_off = j * KB16 + 16 * h;                                  // + 32 * c per 16-d chunk (+ 32*KB16 for lo)
  // transposing V read: lane 4q+p of a 16-lane group addresses row (key) q, columns 4p..4p+3 of the group's 16 columns
  const int v_off = 64 * KB16 + (4 * h + ((lane & 15) >> 2)) * VB16 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  extern __shared__ float rel_dyn[];
  float* const tbl_row = rel_dyn + (wave * 32 + j) * REL_ROW;      // this lane's query
  if (p.rel_key) {      // q . rel_key[r] for this workgroup's queries: the table's rows go through the K planes and the QK MFMAs
    const int nrel = p.rel_left + p.rel_right + 1;
    for (int rt = 0; rt * 32 < nrel; ++rt) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int idx = tid + 256 * l, row = idx >> 4, c4 = idx & 15, rr = rt * 32 + row;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (rr < nrel) a = *reinterpret_cast<const f32x4*>(p.rel_key + (size_t)rr * 64 + c4 * 4);
        bf16x4 hi, lo;
        split4(a, hi, lo);
        *reinterpret_cast<bf16x4*>(smem + row * KB16 + c4 * 8) = hi;
        *reinterpret_cast<bf16x4*>(smem + 32 * KB16 + row * KB16 + c4 * 8) = lo;
      }
      __syncthreads();
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(smem + k_off + 32 * c);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(smem + 32 * KB16 + k_off + 32 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[c], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tbl_row[rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = s[r];
      __syncthreads();
    }
  }

  if (ntiles > 0) {
    load_kv(0);
    store_kv(0);
  }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool has_next = tile + 1 < ntiles;
    if (has_next) load_kv(tile + 1);
    const char* tb = smem + (tile & 1) * ABUF;
    const int key0 = k_lo + tile * 32;
    const bool wave_active = !(p.causal && key0 > q0 + 31) && q0 < p.Sq;      // wave-uniform
    if (wave_active) {
      // ---- S^T = K . Q^T ----
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(tb + k_off + 32 * c);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(tb + 32 * KB16 + k_off + 32 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[c], s, 0, 0, 0);
      }
      if (p.rel_key) add_rel_term(s, tbl_row, key0, q0, qi, h, p.rel_left, p.rel_right);
      // ---- mask (boundary tiles only: wave-uniform test) + online softmax in the log2 domain ----
      const bool need_mask = key0 < kstart || key0 + 32 > kend || (p.causal && key0 + 31 > q0);
      float mx = NEG_BIG;
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = key >= kstart && key < kend && (!p.causal || key <= qi);
          s[r] = ok ? s[r] : NEG_BIG;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = xor32_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
        if (need_mask) pv = s[r] <= -1e29f ? 0.0f : pv;
        s[r] = pv;
        psum += pv;
      }
      l_run = l_run * alpha + psum;
      // the running maximum moves in the first few tiles and then rarely: when no lane's maximum moved (alpha == 1 everywhere,
      // wave-uniform test) the 32 accumulator rescales are skipped -- same bits, a ninth of the loop's VALU work less
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      }
      m_run = m_new;
      // ---- P^T operands: registers 8 t2 .. 8 t2 + 7 are the k-slots of MFMA t2 ----
      bf16x8 ph[2], pl[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const f32x4 a = {s[8 * t2 + 0], s[8 * t2 + 1], s[8 * t2 + 2], s[8 * t2 + 3]};
        const f32x4 c = {s[8 * t2 + 4], s[8 * t2 + 5], s[8 * t2 + 6], s[8 * t2 + 7]};
        bf16x4 ah, al, ch, cl;
        split4(a, ah, al);
        split4(c, ch, cl);
        ph[t2] = __builtin_shufflevector(ah, ch, 0, 1, 2, 3, 4, 5, 6, 7);
        pl[t2] = __builtin_shufflevector(al, cl, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      // ---- O^T += V^T . P^T ----
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* va = tb + v_off + 16 * t2 * VB16 + 64 * t;
          typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
          const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va));
          const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va + 8 * VB16));
          const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va + 32 * VB16));
          const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va + 40 * VB16));
          const bf16x8 vh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[t2], o[t], 0, 0, 0);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[t2], o[t], 0, 0, 0);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[t2], o[t], 0, 0, 0);
        }
    }
    if (has_next) store_kv((tile + 1) & 1);
    __syncthreads();
  }

  const float l_tot = xor32_sum(l_run);
  const float inv_l = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 65);
  static_assert(4 * 32 * 65 * 4 <= 2 * ABUF, "output staging must fit in the K/V tiles");
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      stage[j * 65 + dd] = o[t][r] * inv_l;
    }
  __syncthreads();
  float* ob = p.o ? p.o + (size_t)b * p.o_bs + hd * 64 : nullptr;
  __bf16* const o_hi = static_cast<__bf16*>(p.o_planes);
  __bf16* const o_lo = o_hi ? o_hi + plane_elems(p.B * p.Sq, p.H * 64) : nullptr;
#pragma unroll
  for (int it = 0; it < 32; ++it) {
    const int qq = q0 + it;
    if (qq < p.Sq) {
      const float val = stage[it * 65 + lane];
      if (ob) ob[(size_t)qq * p.o_ts + lane] = val;
      if (o_hi) {
        const __bf16 hi = (__bf16)val;
        const size_t o = plane_index(b * p.Sq + qq, hd * 64 + lane, p.B * p.Sq);
        o_hi[o] = hi;
        o_lo[o] = (__bf16)(val - (float)hi);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Planes variant (DiT attention of s2mel): q / k / v arrive as the split-bf16 planes the qkv GEMM's epilogue wrote (rotary
// embedding already applied), chunk-major [col / 16][row][16]: a 32-key x 16-dim block of one plane is 1 KiB of contiguous
// memory = ONE global_load_lds_dwordx4 wave-instruction.  A K/V tile (32 keys x 64 dims x {K,V} x {hi,lo}) is 16 such pieces,
// four per wave; tiles sit in a 3-slot LDS ring and are requested two tiles ahead of the MFMAs that consume them, so the
// HBM / L2 latency of a tile is covered by two tiles of arithmetic (measured before: with every MFMA, split and exp removed the
// register-staged kernel still took 58 % of its time -- one tile of load latency per iteration).
//   LDS image of a piece: row L (32 bytes: 16 dims) holds key (L ^ 4 (c & 1)) of chunk c, its two 16-byte halves swapped when
//   L & 8 (the swizzle of gemm_bf16x3_v2.hip): ds_read_b128 K fragments and ds_read_b64_tr_b16 V fragments are conflict-free.
// The softmax scale rides in the exponent argument (fma), so q needs no scaling (its hi / lo planes are used as they are).
#define ATT_GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// Diagnostic build only (-DATT_TIMING; never in the product library, tests/test_abi.py checks the exported symbols): per-tile
// s_memtime stamps and a per-workgroup (start, end, XCC | HW_ID, loop cycles) record read back by tools/attn_trace.py.
// -DDBG_PAD=bytes pads the LDS allocation to lower the occupancy (2 workgroups per CU at 16 KiB, 1 at 40 KiB) for the same study.
#ifdef ATT_TIMING
__device__ long long att_trace[4 * 4096];
extern "C" int idxtts_debug_att_trace(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(att_trace), sizeof(long long) * n);
}
#endif
constexpr int ATT_SLOT = 16 * 1024;       // 16 pieces of 1 KiB: [K hi, K lo, V hi, V lo] x 4 chunks
constexpr int ATT_NSLOT = 3;
#ifndef DBG_PAD
#define DBG_PAD 0
#endif

__global__ __launch_bounds__(256) void flash_attn_planes_kernel(const AttnPlanesArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[ATT_NSLOT * ATT_SLOT + DBG_PAD];

  // XCD-aware walk: consecutive workgroup ids go to the 8 XCDs round-robin, each with its own L2.  All query blocks of one
  // (batch row, head) pair read the same K / V planes, so they are given to ONE XCD: pair = 8 * (id / 8 / nq) + id % 8.
  const int L = blockIdx.x, nq = p.nqblk;
  const int qx = L >> 3, pair = (qx / nq) * 8 + (L & 7), qblk = qx % nq;
  if (pair >= p.B * p.H) return;      // whole workgroup (uniform)
  const int b = pair / p.H, hd = pair - b * p.H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int q0 = qblk * 128 + wave * 32;
  const int qi = q0 + j;
  const int kend = p.kend ? min(p.kend[b], p.T) : p.T;
  const int ntiles = (kend + 31) >> 5;
#ifdef ATT_TIMING
  const long long r_kernel = __builtin_amdgcn_s_memrealtime();
#endif
  const __bf16* const hi = static_cast<const __bf16*>(p.planes);
  const __bf16* const lo = hi + plane_elems(p.Mrows, p.ncols);
  const size_t rows16 = (size_t)p.Mrows * 16;

  // Q^T fragments (B operand): lane (query j, half h) holds Q[qi][16c + 8h + e], e = 0..7
  bf16x8 qh[4], ql[4];
  {
    const size_t qrow = (size_t)b * p.T + p.q_row0 + min(qi, p.Sq - 1);
    const size_t base = (size_t)((p.q_col >> 4) + hd * 4) * rows16 + qrow * 16 + 8 * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qh[c] = *reinterpret_cast<const bf16x8*>(hi + base + c * rows16);
      ql[c] = *reinterpret_cast<const bf16x8*>(lo + base + c * rows16);
    }
  }
  // The fragments are consumed here, before the first tile DMA is issued: the compiler places its own wait for the loads now and
  // knows them complete inside the loop.  (With only a hand-written s_waitcnt it kept them "pending" in its counter model and,
  // since DMAs are issued after them, drained vmcnt(0) before the first MFMA of EVERY tile -- the prefetch ring was worth nothing.)
#pragma unroll
  for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(qh[c]), "+v"(ql[c]));

  // DMA role: wave 0 K hi, 1 K lo, 2 V hi, 3 V lo; lane = (LDS row lrow, LDS half unit) of each of the wave's four chunk pieces.
  // Addresses: a wave-uniform 64-bit base per chunk (scalar registers) + a 32-bit per-lane byte offset, two offsets per tile (even
  // and odd chunks differ in the row permutation only) -- the per-tile address arithmetic is six vector instructions.
  const int lrow = lane >> 1;
  const unsigned half_b = (unsigned)(((lane & 1) ^ ((lrow >> 3) & 1)) * 16);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const char* const src0 = reinterpret_cast<const char*>(((wave_s & 1) ? lo : hi) + (size_t)(((wave_s < 2 ? p.k_col : p.v_col) >> 4) + hd * 4) * rows16);
  const size_t chunk_bytes = rows16 * sizeof(__bf16);
  const unsigned row_e = (unsigned)(b * p.T + lrow), row_o = (unsigned)(b * p.T + (lrow ^ 4));
  const unsigned last_row = (unsigned)p.Mrows - 1;
  auto issue_tile = [&](int tile) {
    char* dst = smem + (tile % ATT_NSLOT) * ATT_SLOT + wave_s * 4096;
    // keys past the row's end: finite rows of a neighbour, masked below
    const unsigned off_e = min(row_e + (unsigned)tile * 32u, last_row) * 32u + half_b;
    const unsigned off_o = min(row_o + (unsigned)tile * 32u, last_row) * 32u + half_b;
#pragma unroll
    for (int c = 0; c < 4; ++c) ATT_GLDS16(src0 + c * chunk_bytes + ((c & 1) ? off_o : off_e), dst + c * 1024);
  };

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = NEG_BIG, l_run = 0.0f;
  const float sc = p.scale * 1.4426950408889634f;

  // fragment read offsets inside a slot
  const int kL0 = j, kL1 = j ^ 4;
  const int k_off0 = kL0 * 32 + ((h ^ ((kL0 >> 3) & 1)) << 4);      // even chunks
  const int k_off1 = kL1 * 32 + ((h ^ ((kL1 >> 3) & 1)) << 4);      // odd chunks
  const int g1 = (lane >> 4) & 1, vq = (lane & 15) >> 2, vp = lane & 3;
  // transposing V read: lane (key 4h + vq of the 8-key group, columns 4 vp.. of chunk 2t + g1); +8 keys flips the swizzle
  const int vL = (4 * h + vq) ^ (4 * g1);
  const int v_off_a = 2 * 4096 + g1 * 1024 + vL * 32 + ((vp >> 1) << 4) + ((vp & 1) << 3);            // keys 16 t2 + [0, 8): L & 8 == 0
  const int v_off_b = 2 * 4096 + g1 * 1024 + (vL + 8) * 32 + (((vp >> 1) ^ 1) << 4) + ((vp & 1) << 3);   // keys 16 t2 + [8, 16)

#ifdef ATT_TIMING
  long long tacc[5] = {0, 0, 0, 0, 0};
#define ATT_T(i) { const long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tprev; tprev = tn; }
  long long tprev = __builtin_readcyclecounter();
  const long long t_begin = tprev, r_begin = __builtin_amdgcn_s_memrealtime();
#else
#define ATT_T(i)
#endif
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  if (ntiles > 0) issue_tile(0);
  if (ntiles > 1) issue_tile(1);
  for (int tile = 0; tile < ntiles; ++tile) {
    // tile `tile` has landed: this wave's four pieces by the counted wait (a younger tile may stay in flight), everyone's by the
    // barrier -- which also says every wave is done reading the slot of tile - 1, the one tile + 2 is about to overwrite
    if (tile + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ATT_T(0)
    if (tile + 2 < ntiles) issue_tile(tile + 2);
    ATT_T(1)
    const char* tb = smem + (tile % ATT_NSLOT) * ATT_SLOT;
    const unsigned slot_lds = smem_lds + (tile % ATT_NSLOT) * ATT_SLOT;
    const int key0 = tile * 32;
    if (q0 < p.Sq) {      // wave-uniform
      // ---- S^T = K . Q^T ----
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ko = (c & 1) ? k_off1 : k_off0;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(tb + c * 1024 + ko);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(tb + 4096 + c * 1024 + ko);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[c], s, 0, 0, 0);
      }
      ATT_T(2)
      // V fragments: requested now, behind the QK MFMAs, so that they have landed by the time the softmax is through.
        const unsigned va = slot_lds + v_off_a, vb = slot_lds + v_off_b;
        bf16x4 vf[2][2][4];      // [t2][t][hi a, hi b, lo a, lo b]
#define ATT_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define ATT_TR4(t2, t)                                       \
        ATT_TR(vf[t2][t][0], va, 512 * t2 + 2048 * t);        \
        ATT_TR(vf[t2][t][1], vb, 512 * t2 + 2048 * t);        \
        ATT_TR(vf[t2][t][2], va, 512 * t2 + 2048 * t + 4096); \
        ATT_TR(vf[t2][t][3], vb, 512 * t2 + 2048 * t + 4096);
        ATT_TR4(0, 0) ATT_TR4(0, 1) ATT_TR4(1, 0) ATT_TR4(1, 1)
#undef ATT_TR4
#undef ATT_TR
      // ---- mask (last tile only) + online softmax in the log2 domain, scale folded into the exponent ----
      if (key0 + 32 > kend) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          s[r] = key < kend ? s[r] : NEG_BIG;
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = xor32_max(mx);
      const float m_new = fmaxf(m_run, mx * sc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_new));      // masked: s = -1e30 -> 0
        s[r] = pv;
        psum += pv;
      }
      l_run = l_run * alpha + psum;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      }
      m_run = m_new;
      bf16x8 ph[2], pl[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const f32x4 a = {s[8 * t2 + 0], s[8 * t2 + 1], s[8 * t2 + 2], s[8 * t2 + 3]};
        const f32x4 c = {s[8 * t2 + 4], s[8 * t2 + 5], s[8 * t2 + 6], s[8 * t2 + 7]};
        bf16x4 ah, al, ch, cl;
        split4(a, ah, al);
        split4(c, ch, cl);
        ph[t2] = __builtin_shufflevector(ah, ch, 0, 1, 2, 3, 4, 5, 6, 7);
        pl[t2] = __builtin_shufflevector(al, cl, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      ATT_T(3)
      // ---- O^T += V^T . P^T ----
      // The transposing reads are issued by hand: the compiler models the ds_read_tr intrinsic as an LDS access that may collide with
      // the LDS-DMA writes in flight and drains vmcnt(0) before it -- the tile just requested would be waited for in every iteration.
      // (Its own lgkmcnt bookkeeping does not see these reads; extra outstanding LDS reads only make its counted waits stricter.)
      {
#define ATT_PV(t2, t)                                                                              \
        {                                                                                          \
          const bf16x8 vh = __builtin_shufflevector(vf[t2][t][0], vf[t2][t][1], 0, 1, 2, 3, 4, 5, 6, 7); \
          const bf16x8 vl = __builtin_shufflevector(vf[t2][t][2], vf[t2][t][3], 0, 1, 2, 3, 4, 5, 6, 7); \
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[t2], o[t], 0, 0, 0);                \
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[t2], o[t], 0, 0, 0);                \
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[t2], o[t], 0, 0, 0);                \
        }
#define ATT_WAIT4(n, t2, t) \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(vf[t2][t][0]), "+v"(vf[t2][t][1]), "+v"(vf[t2][t][2]), "+v"(vf[t2][t][3])::"memory")
        ATT_WAIT4(12, 0, 0); ATT_PV(0, 0)
        ATT_WAIT4(8, 0, 1);  ATT_PV(0, 1)
        ATT_WAIT4(4, 1, 0);  ATT_PV(1, 0)
        ATT_WAIT4(0, 1, 1);  ATT_PV(1, 1)
#undef ATT_WAIT4
#undef ATT_PV
      }
      ATT_T(4)
    }
  }
#ifdef ATT_TIMING
  const long long r_loop_end = __builtin_amdgcn_s_memrealtime();
  long long tsave[5] = {tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]};
  const long long t_loop = (long long)__builtin_readcyclecounter() - t_begin;
#endif

  const float l_tot = xor32_sum(l_run);
  const float inv_l = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                            // every wave is done with the ring: reuse it as the transpose buffer
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 65);
  static_assert(4 * 32 * 65 * 4 <= ATT_NSLOT * ATT_SLOT, "output staging must fit in the ring");
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      stage[j * 65 + dd] = o[t][r] * inv_l;
    }
  __syncthreads();
  float* ob = p.o ? p.o + (size_t)b * p.o_bs + hd * 64 : nullptr;
  __bf16* const o_hi = static_cast<__bf16*>(p.o_planes);
  __bf16* const o_lo = o_hi ? o_hi + plane_elems(p.B * p.Sq, p.H * 64) : nullptr;
#pragma unroll
  for (int it = 0; it < 32; ++it) {
    const int qq = q0 + it;
    if (qq < p.Sq) {
      const float val = stage[it * 65 + lane];
      if (ob) ob[(size_t)qq * p.o_ts + lane] = val;
      if (o_hi) {
        const __bf16 vhi = (__bf16)val;
        const size_t oo = plane_index(b * p.Sq + qq, hd * 64 + lane, p.B * p.Sq);
        o_hi[oo] = vhi;
        o_lo[oo] = (__bf16)(val - (float)vhi);
      }
    }
  }
#ifdef ATT_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long r_end = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    const int wg = blockIdx.x;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (wg < 4096) {
      att_trace[4 * wg] = r_kernel; att_trace[4 * wg + 1] = r_end; att_trace[4 * wg + 2] = ((long long)xcc << 32) | hwid;
      att_trace[4 * wg + 3] = t_loop + tsave[0] * 0;
    }
  }
#endif
}

int flash_attn_planes_forward(const AttnPlanesArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.H == 0 || a.Sq == 0) return 0;
  IDX_CHECK(a.planes && (a.o || a.o_planes), "null pointer");
  IDX_CHECK(a.Mrows > 0 && a.T > 0 && (long long)a.B * a.T <= a.Mrows && a.q_row0 >= 0 && a.q_row0 + a.Sq <= a.T, "rows");
  IDX_CHECK((a.ncols & 15) == 0 && (a.q_col & 15) == 0 && (a.k_col & 15) == 0 && (a.v_col & 15) == 0, "columns must be multiples of 16");
  IDX_CHECK(std::max(a.q_col, std::max(a.k_col, a.v_col)) + a.H * 64 <= a.ncols, "columns out of range");
  IDX_CHECK((reinterpret_cast<uintptr_t>(a.planes) & 15) == 0, "planes must be 16-byte aligned");
  IDX_CHECK(a.Mrows < (1 << 26), "rows: per-lane byte offsets are 32-bit");
  AttnPlanesArgs k = a;
  k.nqblk = cdiv(a.Sq, 128);
  const long long pairs8 = (long long)cdiv(a.B * a.H, 8) * 8;
  IDX_CHECK(pairs8 * k.nqblk < (1ll << 31), "grid size");
  dim3 grid((unsigned)(pairs8 * k.nqblk));
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.T * 64;
  const double bytes = 4.0 * a.B * a.H * 64.0 * (2.0 * a.Sq + 2.0 * a.T);
  static const int cat = prof_register("flash_attn_planes_kernel");
  ProfScope prof(cat, stream, flops, bytes);
  hipLaunchKernelGGL(flash_attn_planes_kernel, grid, dim3(256), 0, stream, k);
  IDX_LAUNCH_CHECK();
  return 0;
}

int flash_attn_forward(const AttnArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.H == 0 || a.Sq == 0) return 0;
  IDX_CHECK(a.q && a.k && a.v && (a.o || (a.o_planes && a.split_bf16)), "null pointer");
  IDX_CHECK(a.head_dim == 64, "head_dim must be 64");
  IDX_CHECK((a.q_ts & 3) == 0 && (a.k_ts & 3) == 0 && (a.v_ts & 3) == 0 && (a.q_bs & 3) == 0 && (a.k_bs & 3) == 0 && (a.v_bs & 3) == 0,
            "q/k/v strides must be multiples of 4 floats");
  dim3 grid(cdiv(a.Sq, 128), a.H, a.B);
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.Sk * 64 * (a.causal ? 0.5 : 1.0);
  const double bytes = 4.0 * a.B * a.H * 64.0 * (2.0 * a.Sq + 2.0 * a.Sk);
  static const int cat_x3 = prof_register("flash_attn_bf16x3_kernel"), cat_f32 = prof_register("flash_attn_f32_kernel");
  ProfScope prof(a.split_bf16 ? cat_x3 : cat_f32, stream, flops, bytes);
  size_t dyn = 0;
  if (a.rel_key) {
    IDX_CHECK(a.rel_left >= 0 && a.rel_right >= 0 && a.rel_left + a.rel_right + 1 <= REL_MAX && !a.causal && a.Sq == a.Sk && !a.kstart,
              "relative_key: non-causal self-attention, at most 96 distances");
    IDX_CHECK((reinterpret_cast<uintptr_t>(a.rel_key) & 15) == 0, "rel_key must be 16-byte aligned");
    dyn = REL_LDS;
    static bool attr_set = false;
    if (!attr_set) {
      IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, REL_LDS));
      IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, REL_LDS));
      attr_set = true;
    }
  }
  if (a.split_bf16) hipLaunchKernelGGL(flash_attn_bf16x3_kernel, grid, dim3(256), dyn, stream, a);
  else hipLaunchKernelGGL(flash_attn_f32_kernel, grid, dim3(256), dyn, stream, a);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
