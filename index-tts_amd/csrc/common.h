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
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits): the operand form of every split-bf16 kernel.
// Four elements -> two packed 8-byte groups; 10 VALU instructions (2 v_cvt_pk_bf16_f32 per plane, packed subtract), where
// the vector-convert idiom costs 16.
typedef __bf16 idx_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16_x4(const f32x4 v, idx_bf16x4& hi, idx_bf16x4& lo) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v[0], v[1]}, b2));
  const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v[2], v[3]}, b2));
  const f2 r0 = f2{v[0], v[1]} - f2{__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h0 & 0xffff0000u)};
  const f2 r1 = f2{v[2], v[3]} - f2{__builtin_bit_cast(float, h1 << 16), __builtin_bit_cast(float, h1 & 0xffff0000u)};
  const unsigned l0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, b2));
  const unsigned l1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, b2));
  hi = __builtin_bit_cast(idx_bf16x4, u2{h0, h1});
  lo = __builtin_bit_cast(idx_bf16x4, u2{l0, l1});
}

// Split-bf16 GEMM operands (gemm_bf16x3_v2.hip): activation x[rows][K] fp32 as two bf16 planes (hi, lo = x - hi), each
// [K/16][rows][16]: a 16-k chunk of all rows is one contiguous run.  Producers may write them directly.
__host__ __device__ static inline size_t plane_index(int row, int k, int rows) { return ((size_t)(k >> 4) * rows + row) * 16 + (k & 15); }
__host__ __device__ static inline size_t plane_elems(int rows, int K) { return (size_t)((K + 15) / 16) * rows * 16; }    // per plane; lo plane follows hi
static inline size_t frag_image_floats(int rows, int K) { return (size_t)((rows + 15) / 16) * ((K + 15) / 16) * 256; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace idxtts
