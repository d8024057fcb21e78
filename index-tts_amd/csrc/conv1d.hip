// Implicit-GEMM Conv1d for channels-first activations on the gfx950 fp32 matrix core.
//
// Replaces (for the vocoder and the s2mel convolutions) what the reference leaves to
// torch.nn.Conv1d / ConvTranspose1d: bigvgan.py:362,367,136-139,379; wavenet.py:149,161;
// length_regulator.py:51,61.
//
//   y[b][m][t] = bias[m] + sum_{ci,k} W[m][ci][k] * x[b][ci][t + k*dil - pad_left]
//
// GEMM view per batch row: M = output channels, N = time, K = Cin*taps.  One 256-thread
// workgroup (4 waves) owns a BM x BN output tile; per K-chunk of 16 input channels the
// x tile [16][BN + (K-1)*dil] is staged ONCE in LDS and every tap reads it at a shifted
// column (no im2col in HBM); per (chunk, tap) a BM x 16 weight tile is streamed through a
// second LDS double buffer.  MFMA: v_mfma_f32_32x32x2_f32, exact fp32 (parity with the
// reference's fp32 vocoder).  Weight sub-tiles are pre-packed so a wave's A fragment is one
// linear ds_read_b128 (conflict-free), and the K order inside a chunk is permuted
// (lane-half h, step s -> k = 8g + 4h + s) identically for A and B.
//
// Workgroup -> tile map is XCD-aware: blockIdx round-robins over the 8 XCDs, so XCD x takes
// the n-tiles n = x (mod 8) and walks them m-tile-major; the 32 CUs of an XCD then stream
// the SAME weight slice through their shared 4 MiB L2 while their x tiles differ.
#include "conv1d.h"
#include "conv_epilogue.h"
#include "gemm.h"
#include <string>

#include "prof.h"

namespace idxtts {

// ----------------------------------------------------------------------------------------
// host-side packing
// ----------------------------------------------------------------------------------------
void pack_conv1d(float* dst, const float* w, int Cout, int Cin, int K) {
  const int MT = cdiv(Cout, CONV_MT), NC = cdiv(Cin, CONV_KC);
  for (int mt = 0; mt < MT; ++mt)
    for (int c = 0; c < NC; ++c)
      for (int k = 0; k < K; ++k) {
        float* sub = dst + (((size_t)mt * NC + c) * K + k) * CONV_SUB;
        for (int g = 0; g < 2; ++g)
          for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 32; ++i)
              for (int e = 0; e < 4; ++e) {
                const int co = mt * 32 + i, ci = c * 16 + 8 * g + 4 * h + e;
                sub[((g * 2 + h) * 32 + i) * 4 + e] =
                    (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * K + k] : 0.0f;
              }
      }
}

void pack_conv_transpose1d(float* dst, const float* w, int Cin, int Cout, int Kt, int u) {
  // y[u*s + r] = sum_i x[i] * Wt[ci][co][u*(s-i) + r + p],  p = (Kt-u)/2.  With taps over
  // i = s + tap - 1 (tap = 0,1,2): k = u*(1-tap) + r + p, kept when 0 <= k < Kt.
  const int p = (Kt - u) / 2;
  const int M = Cout * u;
  const int MT = cdiv(M, CONV_MT), NC = cdiv(Cin, CONV_KC);
  for (int mt = 0; mt < MT; ++mt)
    for (int c = 0; c < NC; ++c)
      for (int tap = 0; tap < 3; ++tap) {
        float* sub = dst + (((size_t)mt * NC + c) * 3 + tap) * CONV_SUB;
        for (int g = 0; g < 2; ++g)
          for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 32; ++i)
              for (int e = 0; e < 4; ++e) {
                const int m = mt * 32 + i, ci = c * 16 + 8 * g + 4 * h + e;
                float v = 0.0f;
                if (m < M && ci < Cin) {
                  const int co = m / u, r = m % u;
                  const int k = u * (1 - tap) + r + p;
                  if (k >= 0 && k < Kt) v = w[((size_t)ci * Cout + co) * Kt + k];
                }
                sub[((g * 2 + h) * 32 + i) * 4 + e] = v;
              }
      }
}

// ----------------------------------------------------------------------------------------
// kernel
// ----------------------------------------------------------------------------------------
struct ConvKP {
  const float* x;
  const float* wp;
  const float* bias;
  const float* res;
  float* y;
  int Cin, M, T;
  int K, dil, pad_left, pad_mode;
  int nchunk, mt32;       // K-chunks, packed 32-row tiles
  int ups_log2;           // 0 = plain conv
  int xt, xs;             // x tile width (BN + (K-1)*dil) and LDS row stride
  int ntiles_row, ntiles, nt8;
  float scale;
  int accum;
  const int* lens; int len_mul_out;
};

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256, (TM == 3 ? 2 : 3)) void conv1d_mfma_kernel(const ConvKP p) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int NSUB = TM * WGM;
  constexpr int NXR = (BN + CONV_MAX_HALO + 63) / 64;   // x-tile loads per lane per row
  constexpr int NW4 = NSUB * (CONV_SUB / 4);             // float4s per weight tile
  constexpr int NWL = (NW4 + 255) / 256;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                          // [2][NSUB*512]
  float* Xs = smem + 2 * NSUB * CONV_SUB;    // [2][16*xs]

  // ---- XCD-aware tile assignment ----
  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  const int m_blk = q / p.nt8;
  const int n_idx = (q - m_blk * p.nt8) * 8 + xcd;
  if (n_idx >= p.ntiles) return;
  const int b = n_idx / p.ntiles_row;
  const int t0 = (n_idx - b * p.ntiles_row) * BN;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  const int T = p.T, XS = p.xs, XT = p.xt;
  const float* xrow_base = p.x + (size_t)b * p.Cin * T;

  float xr[4][NXR];
  f32x4 wr[NWL];

  auto load_x = [&](int chunk) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int ci = chunk * CONV_KC + wave * 4 + rr;
      const float* src = xrow_base + (size_t)ci * T;
#pragma unroll
      for (int cc = 0; cc < NXR; ++cc) {
        const int c = lane + 64 * cc;
        int t = t0 - p.pad_left + c;
        if (p.pad_mode == PAD_REFLECT) {
          t = t < 0 ? -t : t;
          t = t >= T ? 2 * (T - 1) - t : t;
        }
        float v = 0.0f;
        if (c < XT && ci < p.Cin && t >= 0 && t < T) v = src[t];
        xr[rr][cc] = v;
      }
    }
  };
  auto store_x = [&](int buf) {
    float* dst = Xs + buf * 16 * XS;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int cc = 0; cc < NXR; ++cc) {
        const int c = lane + 64 * cc;
        if (c < XT) dst[(wave * 4 + rr) * XS + c] = xr[rr][cc];
      }
  };
  auto load_w = [&](int chunk, int tap) {
#pragma unroll
    for (int l = 0; l < NWL; ++l) {
      const int idx = tid + l * 256;
      const int sub = idx >> 7, off = idx & 127;
      const int mt = m_blk * NSUB + sub;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NW4 && mt < p.mt32)
        v = *reinterpret_cast<const f32x4*>(p.wp + (((size_t)mt * p.nchunk + chunk) * p.K + tap) * CONV_SUB + off * 4);
      wr[l] = v;
    }
  };
  auto store_w = [&](int buf) {
    float* dst = Ws + buf * NSUB * CONV_SUB;
#pragma unroll
    for (int l = 0; l < NWL; ++l) {
      const int idx = tid + l * 256;
      if (idx < NW4) *reinterpret_cast<f32x4*>(dst + idx * 4) = wr[l];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  load_x(0);
  load_w(0, 0);
  store_x(0);
  store_w(0);
  __syncthreads();

  const int total = p.nchunk * p.K;
  int chunk = 0, tap = 0;
  for (int it = 0; it < total; ++it) {
    int nchunk_i = chunk, ntap = tap + 1;
    if (ntap == p.K) { ntap = 0; nchunk_i = chunk + 1; }
    const bool has_next = it + 1 < total;
    const bool next_x = has_next && ntap == 0;
    if (has_next) load_w(nchunk_i, ntap);
    if (next_x) load_x(nchunk_i);

    {
      const float* xb = Xs + (chunk & 1) * 16 * XS;
      const float* wb = Ws + (it & 1) * NSUB * CONV_SUB;
      const int colbase = wn * TN * 32 + j + tap * p.dil;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 a[TM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
          a[mt] = *reinterpret_cast<const f32x4*>(wb + (wm * TM + mt) * CONV_SUB + (g * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float* xrow = xb + (8 * g + 4 * h + s) * XS + colbase;
          float bv[TN];
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) bv[nt] = xrow[nt * 32];
#pragma unroll
          for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], bv[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }

    if (has_next) store_w((it + 1) & 1);
    if (next_x) store_x(nchunk_i & 1);
    __syncthreads();
    chunk = nchunk_i;
    tap = ntap;
  }

  // ---- epilogue: bias, residual, scale, (accumulate), store: operands requested in batches ahead of the stores (conv_epilogue.h) ----
  conv_epilogue<TM, TN>(p, acc, m_blk * BM + wm * TM * 32, t0 + wn * TN * 32, b, T, h, j);
}

template <int TM, int TN, int WGM, int WGN>
static int launch_conv(const ConvWeights& w, const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, NSUB = TM * WGM;
  ConvKP p;
  p.x = a.x; p.wp = w.wp; p.bias = w.bias; p.res = a.res; p.y = a.y;
  p.Cin = w.Cin; p.M = w.M; p.T = a.T;
  p.K = w.K; p.dil = a.dil; p.pad_left = a.pad_left; p.pad_mode = a.pad_mode;
  p.nchunk = w.nchunk; p.mt32 = cdiv(w.M, CONV_MT);
  int ul = 0;
  while ((1 << ul) < w.ups) ++ul;
  IDX_CHECK((1 << ul) == w.ups, "transposed-conv stride must be a power of two");
  p.ups_log2 = ul;
  const int halo = (w.K - 1) * a.dil;
  IDX_CHECK(halo <= CONV_MAX_HALO, "(K-1)*dil exceeds CONV_MAX_HALO");
  p.xt = BN + halo;
  p.xs = (p.xt + 3) & ~3;
  p.ntiles_row = cdiv(a.T, BN);
  p.ntiles = p.ntiles_row * a.B;
  p.nt8 = cdiv(p.ntiles, 8);
  p.scale = a.scale; p.accum = a.accum; p.lens = a.lens; p.len_mul_out = a.len_mul_out;
  const int mblocks = cdiv(w.M, BM);
  const size_t lds = (size_t)(2 * NSUB * CONV_SUB + 2 * 16 * p.xs) * sizeof(float);
  const int64_t grid = (int64_t)8 * mblocks * p.nt8;
  IDX_CHECK(grid > 0 && grid < (1ll << 31), "grid size");
  auto kern = conv1d_mfma_kernel<TM, TN, WGM, WGN>;
  static bool attr_set = false;   // one per template instance
  if (!attr_set) {
    IDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  {
    // algorithmic work: 2*Cout*Cin*taps per output sample (a transposed conv has Kt/u = 2 live taps
    // per output, not the 3 the packed form multiplies); bytes = x + y (+ residual / accumulate) + weights once
    const double cout = (double)(w.M / w.ups), tout = (double)a.T * w.ups * a.B;
    const double taps = w.ups > 1 ? 2.0 : (double)w.K;
    const double flops = 2.0 * cout * w.Cin * taps * tout;
    const double bytes = 4.0 * ((double)a.B * w.Cin * a.T + cout * tout * (1.0 + (a.res ? 1.0 : 0.0) + (a.accum ? 1.0 : 0.0)) +
                                cout * w.Cin * (w.ups > 1 ? 2.0 * w.ups : (double)w.K));
    static const int cat = prof_register(("conv1d_mfma_kernel<" + std::to_string(TM) + ", " + std::to_string(TN) + ", " + std::to_string(WGM) + ", " + std::to_string(WGN) + ">").c_str());
    ProfScope prof(cat, stream, flops, bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
  }
  IDX_LAUNCH_CHECK();
  return 0;
}

int conv1d_forward(const ConvWeights& w, const ConvArgs& a, hipStream_t stream) {
  IDX_CHECK(w.wp && a.x && a.y, "null pointer");
  IDX_CHECK(a.B > 0 && a.T > 0, "empty shape");
  if (a.pad_mode == PAD_REFLECT) IDX_CHECK(a.T > (w.K - 1) * a.dil, "reflect pad needs T > halo");
  if (w.wp16 && get_gemm_mode() == GEMM_BF16X3) return conv1d_bf16x3_forward(w, a, stream);
  if (w.M > 96) return launch_conv<2, 2, 2, 2>(w, a, stream);   // 128 x 128
  if (w.M > 64) return launch_conv<3, 2, 1, 4>(w, a, stream);   //  96 x 256
  if (w.M > 32) return launch_conv<2, 2, 1, 4>(w, a, stream);   //  64 x 256
  return launch_conv<1, 4, 1, 4>(w, a, stream);                 //  32 x 512
}

}  // namespace idxtts
