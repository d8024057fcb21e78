// Skinny GEMM for the autoregressive decode step: Y[rows<=64][N] = X[rows][K] * W[K][N], weights
// streamed ONCE from HBM straight into registers (HBM-bound: 4 bytes per weight, 2*rows flop).
//
// Replaces, per generated token, the four Conv1D projections of every GPT-2 block
// (transformers_gpt2.py:304-355 c_attn/c_proj, 578-592 c_fc/c_proj) and the mel_head Linear
// (model_v2.py:208 lm_head) that the reference runs through HF generate (one token per forward,
// model_v2.py:173-178).
//
// CDNA4 mapping: rows (the utterance batch, padded to 16) sit on the M side of v_mfma_f32_16x16x4_f32,
// 16 output columns on the N side; one WAVE owns a (16-column tile, K-slice) and walks its slice with
// one 1 KiB coalesced global_load_dwordx4 per 16 k (weights pre-packed in the B-fragment order
// [N/16][K/16][lane][4], so a wave reads one contiguous stream), several loads in flight, no LDS
// round trip for the weights ("GEMV / M <= 16: load straight to VGPRs").  The activations (<= 64 x K-slice
// floats) are staged once per workgroup in LDS in A-fragment order.  K is split across workgroups to
// put ~2k waves in flight; the K-slice partial sums go to a [KS][rows][N] slab and are combined, in a
// fixed order, in the prologue of the consumer kernel (rows_norm / decode_attn / sample / the next
// gemv16) -- no atomics, bitwise reproducible.
// Optional fused input transform: x = act(sum_s xpart[s] + xbias) (consumes the previous GEMV's slab,
// e.g. gelu_new(c_fc(x) + b) feeding mlp.c_proj).
#include "gemv16.h"
#include "prof.h"

namespace idxtts {

void pack_gemv16_kn(float* dst, const float* w_kn, int K, int N) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 256;
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
          const int k = c * 16 + 4 * (lane >> 4) + s, n = nt * 16 + (lane & 15);
          sub[lane * 4 + s] = (k < K && n < N) ? w_kn[(size_t)k * N + n] : 0.0f;
        }
    }
}

void pack_gemv16_nk(float* dst, const float* w_nk, int N, int K) {
  const int NT = cdiv(N, 16), KC = cdiv(K, 16);
  for (int nt = 0; nt < NT; ++nt)
    for (int c = 0; c < KC; ++c) {
      float* sub = dst + ((size_t)nt * KC + c) * 256;
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
          const int k = c * 16 + 4 * (lane >> 4) + s, n = nt * 16 + (lane & 15);
          sub[lane * 4 + s] = (k < K && n < N) ? w_nk[(size_t)n * K + k] : 0.0f;
        }
    }
}

__device__ __forceinline__ float gelu_new_f(float v) {
  const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  return 0.5f * v * (1.0f + tanhf(u));
}

struct GemvKP {
  const float* x; int ldx;
  const float* xpart; int xparts; int xpart_rows; int ld_xpart; const float* xbias; int xact;
  const float* wp;
  float* ypart;
  int rows, N, K, kc16, ntiles, chunks_per_slice;
};

template <int MT>
__global__ __launch_bounds__(256) void gemv16_kernel(const GemvKP p) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [chunks][MT][64 lanes][4]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ks = blockIdx.y;
  const int c0 = ks * p.chunks_per_slice;
  const int nch = min(p.chunks_per_slice, p.kc16 - c0);
  const int k0 = c0 * 16, kslice = nch * 16;

  constexpr int UN = 8;   // weight loads in flight per wave (8 KiB)
  const int nt = blockIdx.x * 4 + wave;
  const bool wave_ok = nt < p.ntiles;
  const float* wbase = p.wp + ((size_t)(wave_ok ? nt : 0) * p.kc16 + c0) * 256 + lane * 4;
  // the weights do not depend on the activations: put the first 8 KiB per wave in flight BEFORE staging x, so the
  // two HBM round trips of this (microseconds-short) kernel overlap instead of adding up
  f32x4 wpre[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    wpre[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave_ok && u < nch) wpre[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (size_t)u * 256));
  }

  // ---- stage activations: element (row, k) -> A-fragment slot [(k/16)][row/16][ (k%16)/4 *16 + row%16 ][k%4] ----
  const int q4n = kslice >> 2;
  for (int idx = tid; idx < MT * 16 * q4n; idx += 256) {
    const int row = idx / q4n, q4 = idx - row * q4n;
    const int k = k0 + q4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < p.rows && k < p.K) {
      if (p.xpart) {
        f32x4 a = p.xbias ? *reinterpret_cast<const f32x4*>(p.xbias + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* xp = p.xpart + (size_t)row * p.ld_xpart + k;
        const size_t sst = (size_t)p.xpart_rows * p.ld_xpart;
        int s = 0;
        for (; s + 4 <= p.xparts; s += 4) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(xp + (size_t)(s + 0) * sst), t1 = *reinterpret_cast<const f32x4*>(xp + (size_t)(s + 1) * sst);
          const f32x4 t2 = *reinterpret_cast<const f32x4*>(xp + (size_t)(s + 2) * sst), t3 = *reinterpret_cast<const f32x4*>(xp + (size_t)(s + 3) * sst);
          a = (((a + t0) + t1) + t2) + t3;
        }
        for (; s < p.xparts; ++s) a += *reinterpret_cast<const f32x4*>(xp + (size_t)s * sst);
        if (p.xact == 1) { a[0] = gelu_new_f(a[0]); a[1] = gelu_new_f(a[1]); a[2] = gelu_new_f(a[2]); a[3] = gelu_new_f(a[3]); }
        v = a;
      } else {
        v = *reinterpret_cast<const f32x4*>(p.x + (size_t)row * p.ldx + k);
      }
    }
    const int c = q4 >> 2, kq = q4 & 3;
    *reinterpret_cast<f32x4*>(&xs[(((c * MT + (row >> 4)) * 64) + kq * 16 + (row & 15)) * 4]) = v;
  }
  __syncthreads();

  if (!wave_ok) return;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // second batch in flight while the first is consumed
  f32x4 wnext[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    wnext[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (UN + u < nch) wnext[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (size_t)(UN + u) * 256));
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    if (u < nch) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[((u * MT + mt) * 64 + lane) * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wpre[u][s], acc[mt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    if (UN + u < nch) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[(((UN + u) * MT + mt) * 64 + lane) * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wnext[u][s], acc[mt], 0, 0, 0);
      }
    }
  }
  int c = 2 * UN;
  for (; c + UN <= nch; c += UN) {
    f32x4 w[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (size_t)(c + u) * 256));
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[(((c + u) * MT + mt) * 64 + lane) * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], w[u][s], acc[mt], 0, 0, 0);
      }
  }
  for (; c < nch; ++c) {
    const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (size_t)c * 256));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[((c * MT + mt) * 64 + lane) * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], w[s], acc[mt], 0, 0, 0);
    }
  }
  // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + r
  const int n = nt * 16 + (lane & 15);
  if (n < p.N) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + (lane >> 4) * 4 + r;
        if (row < p.rows) p.ypart[((size_t)ks * p.rows + row) * p.N + n] = acc[mt][r];
      }
  }
}

int gemv16_plan_ksplit(int N, int K) {
  const int ntiles = cdiv(N, 16), kc16 = cdiv(K, 16);
  int ks = (2048 + ntiles - 1) / ntiles;            // ~2k waves in flight
  ks = std::max(1, std::min(ks, kc16 / 4 > 0 ? kc16 / 4 : 1));   // >= 4 chunks (4 KiB) per wave
  const int cps = cdiv(kc16, ks);                   // make it self-consistent: every slice but the last is full
  return cdiv(kc16, cps);
}

int gemv16_forward(const Gemv16Weights& w, const Gemv16Args& a, hipStream_t stream) {
  IDX_CHECK(w.wp && a.ypart && (a.x || a.xpart), "null pointer");
  IDX_CHECK(a.rows > 0 && a.rows <= 64, "1..64 rows");
  IDX_CHECK((w.K & 3) == 0, "K must be a multiple of 4");
  if (a.x) IDX_CHECK((a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "x alignment");
  if (a.xpart) IDX_CHECK((a.ld_xpart & 3) == 0, "xpart alignment");
  GemvKP p;
  p.x = a.x; p.ldx = a.ldx;
  p.xpart = a.xpart; p.xparts = a.xparts; p.xpart_rows = a.xpart_rows; p.ld_xpart = a.ld_xpart; p.xbias = a.xbias; p.xact = a.xact;
  p.wp = w.wp; p.ypart = a.ypart;
  p.rows = a.rows; p.N = w.N; p.K = w.K;
  p.kc16 = cdiv(w.K, 16); p.ntiles = cdiv(w.N, 16);
  const int ks = std::max(1, a.ksplit);
  p.chunks_per_slice = cdiv(p.kc16, ks);
  const int ks_eff = cdiv(p.kc16, p.chunks_per_slice);
  IDX_CHECK(ks_eff == ks, "ksplit must divide the K chunks evenly enough (use gemv16_plan_ksplit)");
  const int MT = cdiv(a.rows, 16);
  const size_t lds = (size_t)p.chunks_per_slice * MT * 256 * sizeof(float);
  IDX_CHECK(lds <= 160 * 1024, "K-slice too large for LDS");
  dim3 grid(cdiv(p.ntiles, 4), ks);
  const double flops = 2.0 * a.rows * (double)w.N * w.K;
  const double bytes = 4.0 * ((double)w.N * w.K + (double)ks * a.rows * w.N + (double)a.rows * w.K * (a.xpart ? a.xparts : 1));
  ProfScope prof(PROF_GEMV16, stream, flops, bytes);
#define LAUNCH(MTV)                                                                                          \
  {                                                                                                          \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv16_kernel<MTV>),                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(gemv16_kernel<MTV>, grid, dim3(256), lds, stream, p);                                 \
  }
  if (MT == 1) LAUNCH(1) else if (MT == 2) LAUNCH(2) else if (MT == 3) LAUNCH(3) else LAUNCH(4)
#undef LAUNCH
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
