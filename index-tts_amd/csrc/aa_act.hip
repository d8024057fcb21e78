// Fused anti-aliased SnakeBeta activation: up x2 (12-tap kaiser-sinc, replicate pad) -> SnakeBeta
// (log-scale alpha/beta) -> down x2 (12-tap, replicate pad 5/6), one pass over HBM.
//
// Replaces the reference's only native kernel, `anti_alias_activation_forward`
// (alias_free_activation/cuda/anti_alias_activation_cuda.cu:43-179, launcher :181-209, entry
// `fwd_cuda` :212-246) and is numerically the TORCH path the CPU reference runs
// (alias_free_activation/torch/act.py:25-30, resample.py:28-37, filter.py:94-101,
// activations.py:113-118) -- i.e. the replicate padding of the down-sampler is applied to the
// ACTIVATED up-sampled signal, exactly as torch does.
//
// Index form (oracle/vocoder.py restates the same):
//   u[2j]   = 2*sum_{q<6} f[2q+1]*x[cl(j+2-q)]      u[2j+1] = 2*sum_{q<6} f[2q]*x[cl(j+3-q)]
//   v[m]    = u[m] + 1/(e^beta + 1e-9) * sin(u[m]*e^alpha)^2
//   out[t]  = sum_{k<12} g[k]*v[cl2(2t+k-5)]
//           = sum_{q<6} g[2q+1]*ve[t+q-2] + g[2q]*vo[t+q-3]      (ve[j]=v[2j], vo[j]=v[2j+1])
//
// CDNA4 mapping: one 256-thread workgroup per (row, 1024-sample tile).  The x tile (+8 halo,
// index-clamped = replicate pad) is staged in LDS with coalesced dword loads; the up-sampled,
// activated signal is split into its even/odd polyphase arrays in LDS so that both FIR passes
// read unit-stride, 16-byte-aligned ds_read_b128 (a stride-2 read of an interleaved array
// would be a 2-way bank conflict on every access); each thread produces 4 adjacent outputs and
// stores them as one 16-byte global store.  HBM traffic = 1 read + 1 write per element.
#include "common.h"
#include "prof.h"

namespace idxtts {

constexpr int AA_TILE = 1016;   // outputs per tile: 254 threads x 4, so that the (AA_TILE + 2 AA_VH) / 4 = 256 groups of the up-sampling phase are ONE
                                // pass of the 256 threads (at 1024 outputs the 258 groups sent wave 0 through the whole phase a second time for two lanes:
                                // an eighth of the kernel's VALU issue, and every other wave waited for it at the barrier)
constexpr int AA_TPW = 4;       // consecutive tiles per workgroup (the next tile's loads overlap the current tile's arithmetic)
constexpr int AA_XH = 8;        // x halo each side (>= 6 needed, 8 keeps float4 alignment)
constexpr int AA_VH = 4;        // polyphase halo each side (>= 3 needed)

// element type of x / y: 0 float32, 1 float16, 2 bfloat16 (the reference kernel dispatches on the input dtype the same way,
// anti_alias_activation_cuda.cu:232-244); filters, alpha and beta are float32 and all arithmetic is float32 -- 16-bit inputs are
// widened on load and the result is rounded once on store
template <int DT> struct AAIo;
template <> struct AAIo<0> {
  typedef float T;
  static __device__ __forceinline__ float ld(const void* p, size_t i) { return static_cast<const float*>(p)[i]; }
  static __device__ __forceinline__ void st(void* p, size_t i, float v) { static_cast<float*>(p)[i] = v; }
};
template <> struct AAIo<1> {
  typedef _Float16 T;
  static __device__ __forceinline__ float ld(const void* p, size_t i) { return (float)static_cast<const _Float16*>(p)[i]; }
  static __device__ __forceinline__ void st(void* p, size_t i, float v) { static_cast<_Float16*>(p)[i] = (_Float16)v; }
};
template <> struct AAIo<2> {
  typedef __bf16 T;
  static __device__ __forceinline__ float ld(const void* p, size_t i) { return (float)static_cast<const __bf16*>(p)[i]; }
  static __device__ __forceinline__ void st(void* p, size_t i, float v) { static_cast<__bf16*>(p)[i] = (__bf16)v; }
};

struct AAParams {
  const void* x;
  void* y;
  const float* up_f;     // [12]
  const float* down_f;   // [12]
  const float* log_alpha;  // [C]
  const float* log_beta;   // [C]
  int C, T;
  const int* lens;       // optional [B]: valid samples of row b = lens[b] * len_mul (<= T); the tail is written as zeros and the
  int len_mul;           // replicate padding is applied at the row's OWN end, as a B = 1 call on the unpadded row would
};

// sin(x)^2: pi-periodic and even, so the argument is reduced to r in [-pi/2, pi/2] with a three-term Cody-Waite split of pi
// (n * 3.140625 is exact for |n| < 2^15) and the sign never matters; degree-11 odd Taylor polynomial on r (|error| < 6e-8, the
// size of libm's own).  A dozen instructions instead of sinf's ~40 with its large-argument path: the kernel was VALU-bound on it.
__device__ __forceinline__ float sin_squared_fast(float x) {      // |x| <= 1e5 (the caller checks a whole group of arguments at once)
  const float n = rintf(x * 0.31830988618379067f);
  float r = fmaf(-n, 3.140625f, x);
  r = fmaf(-n, 9.67502593994140625e-4f, r);
  r = fmaf(-n, 1.509957990978376e-7f, r);
  const float r2 = r * r;
  float q = fmaf(r2, -2.5052108385441720e-8f, 2.7557319223985893e-6f);      // -1/11!, 1/9!
  q = fmaf(r2, q, -1.9841269841269841e-4f);                                   // -1/7!
  q = fmaf(r2, q, 8.3333333333333332e-3f);                                    // 1/5!
  q = fmaf(r2, q, -1.6666666666666666e-1f);                                   // -1/3!
  const float sn = fmaf(r * r2, q, r);
  return sn * sn;
}

__device__ __forceinline__ float sin_squared(float x) {
  if (fabsf(x) > 1.0e5f) { const float s = sinf(x); return s * s; }      // never on audio-range activations
  return sin_squared_fast(x);
}

__device__ __forceinline__ float snake(float u, float a, float inv_b) {
  return u + inv_b * sin_squared(u * a);
}
// eight activations at once: ONE range test for the group (a branch per value kept the eight polynomial chains from being interleaved and
// cost three instructions each); the large-argument path (libm's sinf) is taken for the whole group or not at all -- same values either way
__device__ __forceinline__ void snake8(float (&e)[4], float (&o)[4], const float (&ue)[4], const float (&uo)[4], float a, float inv_b) {
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(ue[i] * a), fabsf(uo[i] * a)));
  if (amax > 1.0e5f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i] = snake(ue[i], a, inv_b); o[i] = snake(uo[i], a, inv_b); }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e[i] = ue[i] + inv_b * sin_squared_fast(ue[i] * a);
      o[i] = uo[i] + inv_b * sin_squared_fast(uo[i] * a);
    }
  }
}

// RAGGED = false is the original single-length kernel (the hot path of equal-length batches); RAGGED = true adds the per-row
// length handling of AAParams::lens.
template <bool RAGGED, int DT>
__global__ __launch_bounds__(256) void aa_act_kernel(const AAParams p) {
  typedef AAIo<DT> IO;
  __shared__ __attribute__((aligned(16))) float xs[AA_TILE + 2 * AA_XH];
  __shared__ __attribute__((aligned(16))) float ve[AA_TILE + 2 * AA_VH];
  __shared__ __attribute__((aligned(16))) float vo[AA_TILE + 2 * AA_VH];

  const int tid = threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  const int Tstride = p.T;                                                     // row stride of the padded tensor
  const int T = RAGGED ? min(p.T, p.lens[b] * p.len_mul) : p.T;               // this row's own length
  const size_t row = ((size_t)b * p.C + c) * Tstride;

  // filters and per-channel constants (wave-uniform -> scalar registers)
  float fe[6], fo[6], ge[6], go[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    fe[q] = 2.0f * p.up_f[2 * q + 1];   // even outputs use odd taps (x2 gain folded in)
    fo[q] = 2.0f * p.up_f[2 * q];
    ge[q] = p.down_f[2 * q + 1];        // taps that hit even v
    go[q] = p.down_f[2 * q];
  }
  const float a = expf(p.log_alpha[c]);
  const float inv_b = 1.0f / (expf(p.log_beta[c]) + 1e-9f);

  // A workgroup walks AA_TPW consecutive tiles of its row; the x tile of tile n + 1 is requested (into registers) before tile
  // n is computed, so the HBM round trip of one tile hides behind the arithmetic of the previous one (a 1024-sample tile is
  // only 4 KiB in + 4 KiB out: one tile per workgroup left the kernel waiting on memory for 74 % of its wave cycles).
  constexpr int NXR = (AA_TILE + 2 * AA_XH + 255) / 256;
  float xr[NXR];
  auto gload = [&](int t0n) {
#pragma unroll
    for (int k = 0; k < NXR; ++k) {
      const int i = tid + 256 * k;
      int t = t0n - AA_XH + i;
      t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
      xr[k] = (i < AA_TILE + 2 * AA_XH && T > 0) ? IO::ld(p.x, row + t) : 0.0f;
    }
  };
  const int tile_first = blockIdx.x * AA_TPW;
  if (!(RAGGED && tile_first * AA_TILE >= T)) gload(tile_first * AA_TILE);
  for (int tt = 0; tt < AA_TPW; ++tt) {
  const int t0 = (tile_first + tt) * AA_TILE;
  if (t0 >= Tstride) break;
  if (RAGGED && t0 >= T) {       // tile entirely in the padding of a shorter row: zeros (and so are all later tiles)
    for (int i = tid; i < AA_TILE && t0 + i < Tstride; i += 256) IO::st(p.y, row + t0 + i, 0.0f);
    continue;
  }
  // ---- phase 1: x tile with replicate (clamped) halo (requested one tile ago) ----
  __syncthreads();               // the previous tile's phase 3 is done with ve / vo (and phase 2 with xs)
#pragma unroll
  for (int k = 0; k < NXR; ++k) {
    const int i = tid + 256 * k;
    if (i < AA_TILE + 2 * AA_XH) xs[i] = xr[k];
  }
  __syncthreads();
  {
    const int t0n = t0 + AA_TILE;
    if (tt + 1 < AA_TPW && t0n < Tstride && !(RAGGED && t0n >= T)) gload(t0n);
  }

  // ---- phase 2: polyphase up-sample + SnakeBeta into ve/vo (local index = j - (t0 - AA_VH)) ----
  const int j0 = t0 - AA_VH;
  for (int G = tid; G < (AA_TILE + 2 * AA_VH) / 4; G += 256) {
    const int jg = j0 + 4 * G;
    float e[4], o[4];
    if (jg >= 0 && jg + 3 < T) {
      // x[j-3 .. j+6] for the 4 j's = xs[4G+1 .. 4G+10]
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&xs[4 * G]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&xs[4 * G + 4]);
      const f32x4 x2 = *reinterpret_cast<const f32x4*>(&xs[4 * G + 8]);
      const float w[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
      float ue4[4], uo4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // j = jg+i ; x[j+2-q] = w[i+6-q] ; x[j+3-q] = w[i+7-q]
        float ue = 0.f, uo = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          ue = fmaf(fe[q], w[i + 6 - q], ue);
          uo = fmaf(fo[q], w[i + 7 - q], uo);
        }
        ue4[i] = ue;
        uo4[i] = uo;
      }
      snake8(e, o, ue4, uo4, a, inv_b);
    } else {
      // sequence edge: v is replicate-padded in ITS index space (m clamped to [0, 2T-1])
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = jg + i;
        const int jj = j < 0 ? 0 : (j > T - 1 ? T - 1 : j);
        const int base = jj - (t0 - AA_XH);           // xs index of x[jj]
        float ue = 0.f, uo = 0.f;
        if (base - 3 >= 0 && base + 3 < AA_TILE + 2 * AA_XH) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            ue = fmaf(fe[q], xs[base + 2 - q], ue);
            uo = fmaf(fo[q], xs[base + 3 - q], uo);
          }
        }
        float ev = snake(ue, a, inv_b), ov = snake(uo, a, inv_b);
        if (j < 0) ov = ev;          // v[clamp(2j+1)] = v[0]
        if (j > T - 1) ev = ov;      // v[clamp(2j)]   = v[2T-1]
        e[i] = ev;
        o[i] = ov;
      }
    }
    *reinterpret_cast<f32x4*>(&ve[4 * G]) = f32x4{e[0], e[1], e[2], e[3]};
    *reinterpret_cast<f32x4*>(&vo[4 * G]) = f32x4{o[0], o[1], o[2], o[3]};
  }
  __syncthreads();

  // ---- phase 3: polyphase down-sample, 4 outputs per thread ----
  const int t = t0 + 4 * tid;
  if (4 * tid >= AA_TILE) continue;      // (the tile's last two threads have no outputs; the loop top's barrier is reached all the same)
  if (t >= Tstride) continue;    // (no barrier is skipped: the next tile of this row starts beyond Tstride too -> break above)
  if (RAGGED && t >= T) {        // padding of a shorter row inside a partly valid tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (t + i < Tstride) IO::st(p.y, row + t + i, 0.0f);
    continue;                    // the next tile lies wholly in the padding: no barrier there either
  }
  float ew[12], ow[12];
  {
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(&ve[4 * tid]);
    const f32x4 e1 = *reinterpret_cast<const f32x4*>(&ve[4 * tid + 4]);
    const f32x4 e2 = *reinterpret_cast<const f32x4*>(&ve[4 * tid + 8]);
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(&vo[4 * tid]);
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(&vo[4 * tid + 4]);
    const f32x4 o2 = *reinterpret_cast<const f32x4*>(&vo[4 * tid + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ew[i] = e0[i]; ew[4 + i] = e1[i]; ew[8 + i] = e2[i];
      ow[i] = o0[i]; ow[4 + i] = o1[i]; ow[8 + i] = o2[i];
    }
  }
  float out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // local index of ve[t+i+q-2] = (t+i+q-2) - j0 = 4*tid + i + q + 2 ; vo[t+i+q-3] -> 4*tid + i + q + 1
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      acc = fmaf(go[q], ow[i + q + 1], acc);
      acc = fmaf(ge[q], ew[i + q + 2], acc);
    }
    out[i] = acc;
  }
  if (DT == 0 && (Tstride & 3) == 0 && (!RAGGED || t + 3 < T)) {
    *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + row + t) = f32x4{out[0], out[1], out[2], out[3]};
  } else if (DT != 0 && (Tstride & 3) == 0 && (!RAGGED || t + 3 < T)) {
    typedef typename IO::T h4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<h4*>(static_cast<typename IO::T*>(p.y) + row + t) = h4{(typename IO::T)out[0], (typename IO::T)out[1], (typename IO::T)out[2], (typename IO::T)out[3]};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (t + i < Tstride) IO::st(p.y, row + t + i, t + i < T ? out[i] : 0.0f);
  }
  }   // tiles of this workgroup
}

int aa_act_forward(void* y, const void* x, const float* up_f, const float* down_f, const float* log_alpha,
                   const float* log_beta, int B, int C, int T, hipStream_t stream, const int* lens, int len_mul, int dtype) {
  if (B == 0 || C == 0 || T == 0) return 0;   // reference: seq_len == 0 -> no-op (.cu:193)
  IDX_CHECK(y && x && up_f && down_f && log_alpha && log_beta, "null pointer");
  IDX_CHECK(y != x, "aa_act is not in-place safe (tile halos)");
  IDX_CHECK(C <= 65535 && B <= 65535, "grid y/z limit");
  AAParams p{x, y, up_f, down_f, log_alpha, log_beta, C, T, lens, len_mul};
  dim3 grid(cdiv(cdiv(T, AA_TILE), AA_TPW), C, B);
  static const int cat = prof_register("aa_act_kernel");
  ProfScope prof(cat, stream, 0.0, 8.0 * B * C * (double)T);   // one read + one write per element
  IDX_CHECK(dtype >= 0 && dtype <= 2, "dtype: 0 float32, 1 float16, 2 bfloat16");
  if (dtype == 0) {
    if (lens) hipLaunchKernelGGL((aa_act_kernel<true, 0>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((aa_act_kernel<false, 0>), grid, dim3(256), 0, stream, p);
  } else {
    IDX_CHECK(!lens, "ragged batches are float32 (the vocoder's own tensors)");
    if (dtype == 1) hipLaunchKernelGGL((aa_act_kernel<false, 1>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((aa_act_kernel<false, 2>), grid, dim3(256), 0, stream, p);
  }
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
