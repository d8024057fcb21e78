// Cold-stream read bandwidth probe for decode-GEMV-shaped launches on MI355X (one wave = `chunks` contiguous 1 KiB loads).
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UN, bool NT>
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ w, float* __restrict__ out, int chunks) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const float* base = w + (size_t)wave * chunks * 256 + lane * 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int c = 0; c < chunks; c += UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      v[u] = f32x4{0, 0, 0, 0};
      if (c + u < chunks) {
        const f32x4* p = reinterpret_cast<const f32x4*>(base + (size_t)(c + u) * 256);
        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u];
  }
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 123.456f) out[wave] = s;     // keep the loads alive, (almost) never taken
}

template <int UN, bool NT>
static int run(const float* pool, size_t pool_floats, float* out, size_t bytes, int threads, int chunks, const char* tag) {
  const size_t floats = bytes / 4;
  const size_t waves = floats / ((size_t)chunks * 256);
  const int wpb = threads / 64;
  const int blocks = (int)(waves / wpb);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 40;
  size_t off = 0;
  for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL((probe<UN, NT>), dim3(blocks), dim3(threads), 0, 0, pool + off, out, chunks); off = (off + floats) % (pool_floats - floats); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL((probe<UN, NT>), dim3(blocks), dim3(threads), 0, 0, pool + off, out, chunks);
    off = (off + floats) % (pool_floats - floats);
  }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%-4s bytes=%6.1fMB threads=%4d chunks=%3d UN=%2d blocks=%5d : %7.2f us/launch  %7.1f GB/s\n", tag, bytes / 1e6, threads, chunks, UN, blocks, us, bytes / us / 1e3);
  return 0;
}

int main() {
  const size_t pool_bytes = (size_t)3 << 30;
  float* pool; float* out;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&out, 64 << 20));
  CK(hipMemset(pool, 0, pool_bytes));
  const size_t pf = pool_bytes / 4;
  for (size_t bytes : {(size_t)6553600, (size_t)19660800, (size_t)26214400, (size_t)104857600}) {
    for (int threads : {256, 1024}) {
      for (int chunks : {5, 20, 80}) {
        if (run<8, true>(pool, pf, out, bytes, threads, chunks, "nt")) return 1;
        if (run<8, false>(pool, pf, out, bytes, threads, chunks, "ld")) return 1;
      }
      if (run<16, false>(pool, pf, out, bytes, threads, 80, "ld")) return 1;
      if (run<4, false>(pool, pf, out, bytes, threads, 20, "ld")) return 1;
    }
  }
  return 0;
}
