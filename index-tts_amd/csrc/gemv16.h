#pragma once
#include <algorithm>

#include "common.h"

namespace idxtts {

// Storage format of a decode weight stream.  The arithmetic is the same fp32 MFMA in every format: a compact format only
// changes what is READ (the decode is bound by the weight stream), the values are widened to fp32 in registers.
//   WFMT_F32  4 B / weight
//   WFMT_BF16 2 B / weight (the fp32 value rounded to nearest-even bf16)
//   WFMT_FP8  1 B / weight: OCP e4m3fn code q and a per-output-channel POWER-OF-TWO scale s, weight = s * q (exact in fp32,
//             so s * sum(x q) == sum(x (s q)) bit for bit: the kernel equals the fp32 kernel run on the dequantised matrix)
enum WeightFormat { WFMT_F32 = 0, WFMT_BF16 = 1, WFMT_FP8 = 2 };
static inline int wfmt_bytes(int fmt) { return fmt == WFMT_FP8 ? 1 : (fmt == WFMT_BF16 ? 2 : 4); }

struct Gemv16Weights {      // packed [ceil(N/16)][ceil(K/16)][64 lanes][4] elements of `fmt`
  const void* wp = nullptr;
  const float* wscale = nullptr;    // WFMT_FP8: [N] power-of-two column scales
  int fmt = WFMT_F32;
  int N = 0, K = 0;
};

static inline size_t gemv16_packed_floats(int N, int K) { return (size_t)cdiv(N, 16) * cdiv(K, 16) * 256; }
void pack_gemv16_kn(float* dst, const float* w_kn, int K, int N);   // HF Conv1D [K][N]
void pack_gemv16_nk(float* dst, const float* w_nk, int N, int K);   // nn.Linear [N][K]

// ---- compact formats (host) ----
float fp8_e4m3_decode(unsigned char code);                 // OCP e4m3fn (bias 7, max 448, 0x7f / 0xff = NaN)
unsigned char fp8_e4m3_encode(float v);                    // round to nearest, ties to even code, saturating, never NaN
float bf16_round(float v);                                 // fp32 -> nearest-even bf16 -> fp32
float fp8_column_scale(float maxabs);                      // 2^ceil(log2(maxabs / 448)) (1 for an all-zero column)
// Round a [K][N] (kn) or [N][K] matrix IN PLACE to the values the format can hold (fp8: per output channel n).
void quantize_matrix(float* w, int K, int N, bool kn, int fmt);
// Re-pack an fp32 stream-order pack (pack_gemv16_*) of an already-quantised matrix into the compact stream; fp8 also needs
// the per-column scales (recomputed from the matrix: scale_out[N]).  Fails (returns 1) if a value is not representable.
int compact_gemv16(void* dst, const float* packed_f32, int N, int K, int fmt, float* scale_out);
int fp8_check_device_decode(hipStream_t stream);           // the device's v_cvt_pk_f32_fp8 equals fp8_e4m3_decode on all codes

// Decode GEMV (gemv_fx.hip): activations as A-fragment images (frag_index, common.h), K split across the waves of one
// workgroup, LayerNorm folded into the weights + epilogue, bias / gelu_new / residual epilogue, final output (no slabs).
struct GemvFXArgs {
  const float* xf = nullptr; int rows = 0;        // fragment images [ceil(rows/16)][ceil(K/16)][64][4], padding rows/k zero
  const float* bias = nullptr;                    // [N]; folded layers: c = ln_b . W + bias
  const float* colsum = nullptr; float ln_eps = 1e-5f;   // folded layers: colsum(diag(ln_g) W); weights = diag(ln_g) W
  int act = 0;                                    // 0 none, 1 gelu_new
  const float* res = nullptr;                     // residual in the layout of y (may alias y)
  float* y = nullptr; int y_frag = 0; int ldy = 0;   // y_frag: fragment images over N, else row-major [rows][ldy]
  int ksb = 1;                                    // > 1 (gemv_fx_ksb): K also split across workgroups.  Without `ksb_counters`:
                                                  // y = raw partial sums [ksb][rows][N], no epilogue operands (gemv_fx_combine adds them).
  float* slab = nullptr;                          // With `ksb_counters` (+ slab [ksb][rows][N]): the LAST workgroup of a column tile to
  unsigned* ksb_counters = nullptr;               // arrive adds the partial sums in slab order (fixed, whoever is last) and runs the normal
                                                  // epilogue into y; counters [ceil(N/16)] start at 0 and are left at 0
  int dbg = 0;                                    // ablation mask for tools/gemv_probe.hip only
};
int gemv_fx_ksb(int N, int K);
int gemv_fx_combine(const float* slab, int ksb, int rows, int N, const float* bias, const float* res, float* y, hipStream_t stream);
void gemv_fx_plan(int N, int K, int rows, int* ntw, int* kw);
// 0 (default): 16 waves x 5 chunks (1024 threads, 256 registers per SIMD); 1: the narrow form, 8 waves x 10 chunks (512 threads, <= 176
// registers per SIMD: what ONE retiring workgroup of the acoustic stage's kernels frees on a CU) for 5..16 rows on bf16 / fp8 streams.
// The two forms split K differently, so their results differ in the last bits: a process-wide choice, made before generating.
void set_decode_geometry(int narrow);
int get_decode_geometry();
int gemv_fx_forward(const Gemv16Weights& w, const GemvFXArgs& a, hipStream_t stream);

}  // namespace idxtts
