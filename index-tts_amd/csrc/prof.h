// Optional per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default: when disabled the hooks cost one predictable branch and record nothing.
#pragma once
#include "common.h"

namespace idxtts {

enum ProfCat {
  PROF_CONV_128x128 = 0, PROF_CONV_96x256, PROF_CONV_64x256, PROF_CONV_32x512,
  PROF_AA_ACT, PROF_CONV_POST,
  PROF_GEMM_TN, PROF_GEMM_BF16X3, PROF_GEMM_BF16X3_256x128, PROF_GEMM_BF16X3_256x256, PROF_FLASH_ATTN, PROF_ROWS_NORM, PROF_GEMV16, PROF_DECODE_ATTN, PROF_SAMPLE, PROF_EMBED, PROF_ELTWISE,
  PROF_NCAT
};

const char* prof_name(int cat);
bool prof_enabled();
// record the start/stop events of one launch of family `cat` doing `flops` algorithmic FLOPs and
// moving `bytes` algorithmic bytes (compulsory HBM traffic: each operand once)
void prof_begin(int cat, hipStream_t stream, double flops, double bytes);
void prof_end(int cat, hipStream_t stream);

struct ProfScope {
  int cat; hipStream_t s; bool on;
  ProfScope(int c, hipStream_t st, double flops, double bytes) : cat(c), s(st), on(prof_enabled()) { if (on) prof_begin(cat, s, flops, bytes); }
  ~ProfScope() { if (on) prof_end(cat, s); }
};

}  // namespace idxtts
