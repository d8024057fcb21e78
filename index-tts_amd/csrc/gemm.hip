// Token-major GEMM on the gfx950 fp32 matrix core: Y[M][N] = epi(X[M][K] * W^T + bias).
//
// Replaces the nn.Linear / HF Conv1D call sites of the hot path:
//   GPT-2 block c_attn / c_proj / c_fc / mlp.c_proj  (indextts/gpt/transformers_gpt2.py:304-355, 578-592)
//   DiT wqkv / wo / w1,w3 / w2 / skip_in_linear      (indextts/s2mel/modules/gpt_fast/model.py:233-234, 270-308, 311-319)
//   DiT cond_projection, cond_x_merge_linear, skip_linear, conv1, res_projection, final_layer, conv2
//                                                    (indextts/s2mel/modules/diffusion_transformer.py:213-252)
//   gpt_layer / content_in_proj                      (commons.py:413, length_regulator.py:88)
//
// 128x128 output tile per 256-thread workgroup (2x2 waves, each 2x2 tiles of v_mfma_f32_32x32x2_f32),
// K stepped 32 at a time through an LDS double buffer with register staging (loads issued before the
// 64-MFMA block, LDS writes after it, one barrier per step).  Both operands are K-contiguous in HBM;
// in LDS both are kept in the MFMA fragment order [g][h][row][4] so every fragment read is one
// ds_read_b128 of 64 x 16 contiguous bytes (conflict-free).  The activation tile gets there through a
// transposing register->LDS write; its 512-byte sub-blocks are padded to 528 B so the 8 lanes that hold
// one 128-byte row segment hit 8 different bank groups.  W is packed once at context creation.
// XCD-aware tile map: XCD x owns the m-tiles == x (mod 8) and walks them n-block-major, so the 32 CUs of
// an XCD share one weight slice in their L2 while streaming different activation rows.
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>

#include "gemm_common.h"
#include "prof.h"

namespace idxtts {

void pack_linear(float* dst, const float* w, int N, int K) {
  const int NT = cdiv(N, 32), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 512;
      for (int g = 0; g < 2; ++g)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 32; ++j)
            for (int e = 0; e < 4; ++e) {
              const int n = nt * 32 + j, k = c * 16 + 8 * g + 4 * h + e;
              sub[((g * 2 + h) * 32 + j) * 4 + e] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.0f;
            }
    }
}

void pack_linear_kn(float* dst, const float* w_kn, int K, int N) {
  const int NT = cdiv(N, 32), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 512;
      for (int g = 0; g < 2; ++g)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 32; ++j)
            for (int e = 0; e < 4; ++e) {
              const int n = nt * 32 + j, k = c * 16 + 8 * g + 4 * h + e;
              sub[((g * 2 + h) * 32 + j) * 4 + e] = (n < N && k < K) ? w_kn[(size_t)k * N + n] : 0.0f;
            }
    }
}

constexpr int XBLK = 132;   // floats per padded [32 rows][4] sub-block (528 B)

// per-(device, stream) scratch of the few-tile launches' K-group combine: [4 KiB arrival counters][partial tiles]
struct SplitKScratch { void* ptr = nullptr; size_t bytes = 0; };
static std::map<std::pair<int, hipStream_t>, SplitKScratch> g_sk;
static std::mutex g_sk_mu;
int gemm_tn_release_stream_scratch(hipStream_t stream) {
  int dev_id = 0;
  IDX_HIP(hipGetDevice(&dev_id));
  SplitKScratch sc;
  {
    std::lock_guard<std::mutex> lock(g_sk_mu);
    auto it = g_sk.find(std::make_pair(dev_id, stream));
    if (it == g_sk.end()) return 0;
    sc = it->second;
    g_sk.erase(it);
  }
  if (sc.ptr) IDX_HIP(hipFreeAsync(sc.ptr, stream));
  return 0;
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmKP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*Xs)[4 * 2 * 4 * XBLK] = reinterpret_cast<float (*)[4 * 2 * 4 * XBLK]>(smem);
  float (*Ws)[4 * 2 * 512] = reinterpret_cast<float (*)[4 * 2 * 512]>(smem + 2 * 4 * 2 * 4 * XBLK);

  // XCD x owns the m-tiles == x (mod 8).  Walk order inside an XCD:
  //   n_fast = 1 (weights fit the 4 MiB L2): all n-blocks of one m-tile back to back -> the activation tile is fetched
  //              from HBM once and W stays L2-resident (DiT / WaveNet shapes: W <= 6 MB, X = 100+ MB)
  //   n_fast = 0 (big W, few rows: GPT prefill): all m-tiles of one n-block back to back -> W streams once per XCD
  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  int bn, bm;
  if (p.direct_map) { bn = L / p.mtiles; bm = L - bn * p.mtiles; }      // few tiles: one per workgroup id, spread over all XCDs
  else if (p.n_fast) { const int bml = q / p.nblocks; bn = q - bml * p.nblocks; bm = bml * 8 + xcd; }
  else { bn = q / p.mt8; bm = (q - bn * p.mt8) * 8 + xcd; }
  if (bm >= p.mtiles) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 xr[4], wr[4];
  // conv mode: (sequence base row, position in sequence) of the 4 activation rows this thread stages
  int seq_base[4], seq_t[4], seq_n[4];     // seq_n: true length of that sequence (reflect pad bounces at ITS end)
  if (p.taps > 1) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int m = bm * 128 + ((tid + 256 * l) >> 3);
      const int sb = m / p.seq_len;
      seq_base[l] = sb * p.seq_len;
      seq_t[l] = m - sb * p.seq_len;
      seq_n[l] = (p.row_len && m < p.M) ? min(p.row_len[sb], p.seq_len) : p.seq_len;
    }
  }
  auto load_tiles = [&](int kstep) {
    int tap = 0, kk0 = kstep * 32;
    if (p.taps > 1) { tap = kk0 / p.kc; kk0 -= tap * p.kc; }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 256 * l;
      // activations: 8 lanes per 128-byte row segment
      const int row = idx >> 3, q8 = idx & 7;
      const int m = bm * 128 + row, k = kstep * 32 + q8 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p.taps > 1) {
        int t = seq_t[l] + tap * p.dil - p.pad_left;
        if (p.pad_mode == 1) { t = t < 0 ? -t : t; t = t >= seq_n[l] ? 2 * (seq_n[l] - 1) - t : t; }
        if (m < p.M && k < p.K && t >= 0 && t < seq_n[l])
          v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(seq_base[l] + t) * p.ldx + kk0 + q8 * 4);
      } else if (m < p.M && k < p.K) {
        v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + k);
      }
      xr[l] = v;
      // weights: 4 KiB contiguous per 32-column tile (two 16-wide chunks)
      const int nt = idx >> 8, off = idx & 255;
      const int ntg = bn * 4 + nt, c16 = kstep * 2 + (off >> 7);
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
      if (ntg * 32 < p.N && c16 < p.kc16)
        u = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)ntg * p.kc16 + c16) * 512 + (off & 127) * 4);
      wr[l] = u;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int idx = tid + 256 * l;
      const int row = idx >> 3, q8 = idx & 7;
      const int mt = row >> 5, i = row & 31, c = q8 >> 2, gh = q8 & 3;
      *reinterpret_cast<f32x4*>(&Xs[buf][((mt * 2 + c) * 4 + gh) * XBLK + i * 4]) = xr[l];
      const int nt = idx >> 8, off = idx & 255;
      *reinterpret_cast<f32x4*>(&Ws[buf][nt * 1024 + off * 4]) = wr[l];
    }
  };

  // acc: the K group in progress; tot: the finished groups, added in group order (see GemmKP::kg)
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.0f; tot[a][b][r] = 0.0f; }

  int ksteps = (p.kc16 + 1) >> 1, ks0 = 0;
  if (p.ksplit > 1) { ks0 = blockIdx.y * p.ksteps_per_split; ksteps = min(ksteps, ks0 + p.ksteps_per_split); }
  int in_group = ks0 % p.kg;
  load_tiles(ks0);
  store_tiles(ks0 & 1);
  __syncthreads();
  for (int ks = ks0; ks < ksteps; ++ks) {
    const bool has_next = ks + 1 < ksteps;
    if (has_next) load_tiles(ks + 1);
    const float* xb = Xs[ks & 1];
    const float* wb = Ws[ks & 1];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t] = *reinterpret_cast<const f32x4*>(xb + (((wm * 2 + t) * 2 + c) * 4 + (g * 2 + h)) * XBLK + j * 4);
          b[t] = *reinterpret_cast<const f32x4*>(wb + (wn * 2 + t) * 1024 + c * 512 + (g * 64 + lane) * 4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
      }
    if (++in_group == p.kg || !has_next) {      // a K group is complete (or the range ends)
      in_group = 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[mt][nt][r] += acc[mt][nt][r]; acc[mt][nt][r] = 0.0f; }
    }
    if (has_next) store_tiles((ks + 1) & 1);
    __syncthreads();
  }

  if (p.sk_cnt) {
    // one K group per workgroup: the partial tile goes to the slab in register order (write-through 16-byte stores: [group][tile][16][256
    // threads]), the last workgroup of the tile to arrive adds the groups in order
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ int s_last;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_slab, 0, p.sk_slab_bytes, 0x00020000);
    const int tile = bm * p.nblocks + bn, ntiles = p.mtiles * p.nblocks;
    const int part_stride = ntiles * 16 * 256 * 16;      // bytes between two groups
    const int mine = (tile * 16 * 256 + tid) * 16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {tot[mt][nt][4 * r4], tot[mt][nt][4 * r4 + 1], tot[mt][nt][4 * r4 + 2], tot[mt][nt][4 * r4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, mine + ((mt * 2 + nt) * 4 + r4) * 4096, (int)blockIdx.y * part_stride, 17);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // acknowledged at the device coherence point before the arrival
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&p.sk_cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)p.ksplit - 1u;
      if (s_last) __hip_atomic_store(&p.sk_cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[mt][nt][r] = 0.0f;
    for (int g = 0; g < p.ksplit; ++g) {
      u32x4 t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, mine + u * 4096, g * part_stride, 16);      // sc1: past the L1
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const f32x4 v = __builtin_bit_cast(f32x4, t[u]);
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[u >> 3][(u >> 2) & 1][4 * (u & 3) + e] += v[e];
      }
    }
    gemm_epilogue(p, tot, bm, bn, wm, wn, h, j);
    return;
  }
  if (p.ksplit > 1) {      // raw partial slab of this K range
    GemmKP q = p;
    q.y = p.y + (size_t)blockIdx.y * p.M * p.ldy;
    q.bias = nullptr; q.res = nullptr; q.row_len = nullptr; q.act = ACT_NONE; q.out_scale = 1.0f;
    gemm_epilogue(q, tot, bm, bn, wm, wn, h, j);
    return;
  }
  gemm_epilogue(p, tot, bm, bn, wm, wn, h, j);
}

int gemm_tn_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream) {
  IDX_CHECK(w.wp && a.x && a.y, "null pointer");
  if (a.M == 0) return 0;
  IDX_CHECK(a.M > 0 && w.N > 0 && w.K > 0, "bad shape");
  IDX_CHECK((w.K & 3) == 0 && (a.ldx & 3) == 0, "K and ldx must be multiples of 4 (16-byte row segments)");
  IDX_CHECK((reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "x must be 16-byte aligned");
  if (a.act == ACT_SWIGLU || a.act == ACT_GATE) IDX_CHECK((w.N & 63) == 0, "paired activations need N (= 2*hidden) to be a multiple of 64");
  if (a.taps > 1) {
    IDX_CHECK(a.seq_len > 0 && a.M % a.seq_len == 0 && w.K % a.taps == 0 && ((w.K / a.taps) & 31) == 0, "conv mode: K/taps must be a multiple of 32 and M a multiple of seq_len");
    if (a.pad_mode == 1) IDX_CHECK(a.seq_len > (a.taps - 1) * a.dil, "reflect pad needs seq_len > halo");
  }
  if (a.row_len) IDX_CHECK(a.seq_len > 0, "row_len needs seq_len");
  GemmKP p;
  IDX_CHECK(a.x && a.y && !a.y_planes && !a.rope, "the fp32 GEMM takes and produces fp32 rows only (no fused rotary)");
  p.x = a.x; p.wp = w.wp; p.bias = w.bias; p.res = a.res; p.y = a.y; p.y_hi = p.y_lo = nullptr; p.rope = nullptr; p.rope_T = 1; p.rope_cols = 0;
  p.M = a.M; p.N = w.N; p.K = w.K; p.ldx = a.ldx; p.ldy = a.ldy; p.ldr = a.ldr;
  p.kc16 = cdiv(w.K, 16);
  p.mtiles = cdiv(a.M, 128);
  p.mt8 = cdiv(p.mtiles, 8);
  p.act = a.act; p.out_scale = a.out_scale;
  p.taps = a.taps; p.kc = w.K / std::max(1, a.taps); p.seq_len = a.seq_len > 0 ? a.seq_len : 1; p.dil = a.dil; p.pad_left = a.pad_left;
  p.pad_mode = a.pad_mode; p.row_len = a.row_len;
  p.ksplit = std::max(1, a.ksplit);
  p.ksteps_per_split = cdiv(cdiv(p.kc16, 2), p.ksplit);
  if (p.ksplit > 1) {
    IDX_CHECK(a.taps <= 1 && a.act == ACT_NONE && !a.res && !a.row_len, "split-K launches produce raw partial slabs");
    IDX_CHECK((p.ksplit - 1) * p.ksteps_per_split < cdiv(p.kc16, 2), "ksplit leaves an empty K range");
  }
  const int nblocks = cdiv(w.N, 128);
  p.nblocks = nblocks;
  p.n_fast = ((double)w.N * w.K * 4.0 <= 8.0 * 1024 * 1024) && ((double)a.M * w.K > (double)w.N * w.K) ? 1 : 0;
  // K groups (GemmKP::kg): a function of K alone, so a row's result never depends on how many rows are computed beside it
  const int ksteps_all = cdiv(p.kc16, 2);
  p.kg = std::max(5, cdiv(ksteps_all, 8));
  const int ngroups = cdiv(ksteps_all, p.kg);
  p.direct_map = 0; p.sk_slab = nullptr; p.sk_cnt = nullptr; p.sk_slab_bytes = 0;
  const int tiles = nblocks * p.mtiles;
  // Few output tiles (a B = 1 prefill, the prompt encoders' projections): one workgroup per tile would leave most CUs idle behind
  // a K loop of dependent global-load -> LDS -> MFMA steps.  One K group per workgroup instead (up to 8 x the workgroups), combined in
  // the kernel by the last arriver.
  if (p.ksplit == 1 && tiles <= 96 && ngroups >= 2) {
    const size_t slab_bytes = (size_t)ngroups * tiles * 65536, cnt_bytes = 4096;
    static_assert(96 * 4 <= 4096, "arrival counters");
    int dev_id = 0;
    IDX_HIP(hipGetDevice(&dev_id));
    SplitKScratch* scp = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_sk_mu);
      scp = &g_sk[std::make_pair(dev_id, stream)];
    }
    if (scp->bytes < cnt_bytes + slab_bytes) {      // stream-ordered (no device-wide synchronisation), grow-only
      if (scp->ptr) IDX_HIP(hipFreeAsync(scp->ptr, stream));
      scp->ptr = nullptr; scp->bytes = 0;
      IDX_HIP(hipMallocAsync(&scp->ptr, cnt_bytes + slab_bytes, stream));
      IDX_HIP(hipMemsetAsync(scp->ptr, 0, cnt_bytes, stream));      // counters start at 0 and every launch leaves them at 0
      scp->bytes = cnt_bytes + slab_bytes;
    }
    p.sk_cnt = static_cast<unsigned*>(scp->ptr);
    p.sk_slab = reinterpret_cast<float*>(static_cast<char*>(scp->ptr) + cnt_bytes);
    p.sk_slab_bytes = (int)slab_bytes;
    p.ksplit = ngroups; p.ksteps_per_split = p.kg;
  }
  if (tiles <= 96) p.direct_map = 1;
  const int64_t grid = p.direct_map ? (int64_t)tiles : (int64_t)8 * nblocks * p.mt8;
  IDX_CHECK(grid < (1ll << 31), "grid size");
  const double flops = 2.0 * a.M * (double)w.N * w.K;
  const double bytes = 4.0 * ((double)a.M * w.K + (double)w.N * w.K + (double)a.M * w.N * (a.res ? 2.0 : 1.0));
  static const int cat = prof_register("gemm_tn_kernel");
  ProfScope prof(cat, stream, flops, bytes);
  constexpr size_t lds = (size_t)(2 * 4 * 2 * 4 * XBLK + 2 * 4 * 2 * 512) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)grid, (unsigned)p.ksplit), dim3(256), lds, stream, p);
  IDX_LAUNCH_CHECK();
  return 0;
}

static int g_gemm_mode = GEMM_BF16X3;
void set_gemm_mode(int mode) { g_gemm_mode = mode == GEMM_F32 ? GEMM_F32 : GEMM_BF16X3; }
int get_gemm_mode() { return g_gemm_mode; }

bool gemm_bf16x3_uses_v2(const LinearWeights& w, const GemmArgs& a);
bool gemm_uses_planes(const LinearWeights& w, const GemmArgs& a) {
  return g_gemm_mode == GEMM_BF16X3 && w.wp16 && a.M >= 256 && gemm_bf16x3_uses_v2(w, a);
}
int gemm_forward(const LinearWeights& w, const GemmArgs& a, hipStream_t stream) {
  if (g_gemm_mode == GEMM_BF16X3 && w.wp16 && a.M >= 256) return gemm_bf16x3_forward(w, a, stream);
  return gemm_tn_forward(w, a, stream);
}

}  // namespace idxtts
