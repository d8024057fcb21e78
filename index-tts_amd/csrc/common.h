// Shared host/device helpers for libidxtts_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace idxtts {

// Thread-local last-error string, read through idxtts_last_error() (never throws across the ABI).
void set_error(const std::string& msg);
int fail(const char* file, int line, const std::string& msg);

#define IDX_FAIL(msg) return ::idxtts::fail(__FILE__, __LINE__, (msg))
#define IDX_CHECK(cond, msg) do { if (!(cond)) IDX_FAIL(std::string("check failed: " #cond " : ") + (msg)); } while (0)
#define IDX_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) \
    IDX_FAIL(std::string(#expr " -> ") + hipGetErrorString(e__)); } while (0)
#define IDX_LAUNCH_CHECK() IDX_HIP(hipGetLastError())

// Decode-step activations are kept as MFMA A-fragment images (v_mfma_f32_16x16x4_f32): [rows/16][K/16][64 lanes][4],
// lane = (k % 16) / 4 * 16 + row % 16, component = k % 4 -- the fragment of a 16-k chunk is one contiguous KiB.
__host__ __device__ static inline size_t frag_index(int row, int k, int kc16) {
  return ((((size_t)(row >> 4) * kc16 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (row & 15)) << 2) + (k & 3);
}
// Split-bf16 GEMM operands (gemm_bf16x3_v2.hip): activation x[rows][K] fp32 as two bf16 planes (hi, lo = x - hi), each
// [K/16][rows][16]: a 16-k chunk of all rows is one contiguous run.  Producers may write them directly.
__host__ __device__ static inline size_t plane_index(int row, int k, int rows) { return ((size_t)(k >> 4) * rows + row) * 16 + (k & 15); }
__host__ __device__ static inline size_t plane_elems(int rows, int K) { return (size_t)((K + 15) / 16) * rows * 16; }    // per plane; lo plane follows hi
static inline size_t frag_image_floats(int rows, int K) { return (size_t)((rows + 15) / 16) * ((K + 15) / 16) * 256; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace idxtts
