// Optional per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default: when disabled the hooks cost one predictable branch and record nothing.
#pragma once
#include "common.h"

namespace idxtts {

// Families are named after the kernel function they time, as rocprofv3 prints it without "void idxtts::" and the argument list
// ("gemm_bf16x3_v2_kernel", "conv1d_bf16x3_kernel<2, 2, 2, 2>", ...): every line of bench.py's per-kernel table matches the rows of
// profiles/rNN_*_kernel_stats.csv whose name contains it.  A launch site registers its family once:
//     static const int cat = prof_register("decode_attn_kernel");
constexpr int PROF_MAX = 128;
int prof_register(const char* kernel_name);   // idempotent, thread-safe

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
