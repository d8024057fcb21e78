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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace idxtts
