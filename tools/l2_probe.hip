// Per-CU load bandwidth by residency level on MI355X: every wave streams 16 B/lane loads over a window that fits the
// per-XCD L2 (2 MiB), the Infinity Cache (64 MiB) or only HBM (2 GiB).
//   hipcc -O3 --offload-arch=gfx950 tools/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UN>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, size_t window_chunks, int iters) {
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  f32x4 acc = {0, 0, 0, 0};
  size_t c = (size_t)gw * UN;
  for (int it = 0; it < iters; ++it) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const f32x4*>(w + ((c + u) % window_chunks) * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u];
    c += (size_t)nw * UN;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[gw] = acc[0];
}

int main() {
  const size_t pool_bytes = (size_t)2 << 30;
  float *pool, *out;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&out, 1 << 22)); CK(hipMemset(pool, 0, pool_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t win : {(size_t)2 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)2 << 30}) {
    for (int wpc : {4, 8, 16}) {            // waves per CU (one workgroup per CU)
      const int iters = 400;
      const size_t chunks = win / 1024;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<8>), dim3(256), dim3(64 * wpc), 0, 0, pool, out, chunks, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = 256.0 * wpc * iters * 8 * 1024;
      printf("window %5zu MiB  waves/CU %2d  : %8.1f GB/s total  %6.1f GB/s/CU  (%5.1f B/clk/CU at 2.4 GHz)\n", win >> 20, wpc, bytes / ms / 1e6,
             bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
    }
  }
  return 0;
}
