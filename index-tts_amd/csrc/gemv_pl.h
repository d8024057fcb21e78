#pragma once
#include "gemv16.h"

namespace idxtts {

// Decode GEMV on the bf16 matrix pipe, for compact (bf16 / fp8) weight streams: Y[rows <= 64][N] = epi(LN?(X)[rows][K] . W[K][N]).
//
// An fp32 activation is carried as THREE bf16 planes x = h + m + l (h = the top 8 significant bits, m the next 8, l the last 8: the
// split is exact), the weights already are bf16 (or fp8, widened exactly), so x . w = h . w + m . w + l . w in three
// v_mfma_f32_16x16x32_bf16 with exact products and fp32 accumulation -- the arithmetic class of the fp32 MFMA GEMV (gemv_fx.hip) at
// 3/16 of its matrix-pipe time, which is what lets ONE launch serve 48-64 rows (three or four coalesced 16-utterance decodes)
// for the cost of one 16-row launch: the weight stream is read once for all of them.
//
// Layouts:
//   weights      [N/16][K/32][64 lanes][8] bf16: B fragments of v_mfma_f32_16x16x32_bf16, lane l holds k = 32 c + 8 (l >> 4) + j, n = 16 t + (l & 15)
//                (fp8: the same order, one byte per weight + [N] power-of-two column scales)
//   activations  fp32 rows [rows][ldx] in HBM (4 bytes per element past the L2, not 6): a workgroup splits the K part it needs into the
//                three planes ONCE, on the way into LDS (A-fragment order: lane l holds row 16 mt + (l & 15), k = 32 c + 8 (l >> 4) + j)
struct Gemv32Weights {
  const void* wp = nullptr;
  const float* wscale = nullptr;    // WFMT_FP8: [N] power-of-two column scales
  int fmt = WFMT_BF16;
  int N = 0, K = 0;
};
static inline size_t gemv32_packed_elems(int N, int K) { return (size_t)cdiv(N, 16) * cdiv(K, 32) * 512; }
// Pack an already-quantised matrix (every value representable in `fmt`) given as [K][N] (kn) or [N][K]; fp8 also returns the per-column
// scales.  Returns 1 if a value is not representable.
int pack_gemv32(void* dst, const float* w, int N, int K, bool kn, int fmt, float* scale_out);

struct GemvPLArgs {
  const float* x = nullptr; int ldx = 0; int rows = 0;  // activations, fp32 rows (16-byte aligned, ldx % 4 == 0)
  const float* bias = nullptr;                          // [N]; folded layers: c = ln_b . W + bias
  const float* colsum = nullptr; float ln_eps = 1e-5f;  // folded LayerNorm: u = colsum(diag(ln_g) W); weights = diag(ln_g) W
  // folded LayerNorm: the row statistics come from the PRODUCER of x as per-16-column partials [stats_tiles][MT * 16][2] = (mean, M2) of each
  // row over the 16 columns of a tile; every workgroup combines them (Chan's update, fixed order) while its weight stream is in flight
  const float* stats_in = nullptr; int stats_tiles = 0;
  int act = 0;                                          // 0 none, 1 gelu_new
  const float* res = nullptr;                           // residual [rows][ldy] (may alias y)
  float* y = nullptr; int ldy = 0;                      // fp32 row-major output
  float* stats_out = nullptr;                           // [N / 16][MT * 16][2] row statistics of the output per column tile, or null
  float* slab = nullptr; unsigned* counters = nullptr;  // K split across workgroups: [kp][N / 16][MT][64][4] partial sums + one arrival
                                                        // counter per workgroup column (0 on entry, left at 0): needed when the plan has K parts
};
// geometry of a launch (column tiles per workgroup; K parts across workgroups) for the given shape
void gemv_pl_plan(int N, int K, int rows, int* ct, int* kparts);
size_t gemv_pl_slab_floats(int N, int K, int rows);
void set_decode_plane_rows(int min_rows);       // include/idxtts.h::idxtts_set_decode_plane_rows
int get_decode_plane_rows();
void gemv_pl_set_ct_override(int ct);      // measurement hook (tools/gemv_pl_probe.hip): 0 = the plan
int gemv_pl_forward(const Gemv32Weights& w, const GemvPLArgs& a, hipStream_t stream);

}  // namespace idxtts
