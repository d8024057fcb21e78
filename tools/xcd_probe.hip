// XCD-partitioning probe for MI355X (8 XCDs x 32 CUs): can two kernels be confined to disjoint XCD sets by letting the
// workgroups that land on the "wrong" XCD exit at once, and what does a decode-GEMV-shaped stream cost on 2 of 8 XCDs,
// alone and beside a compute hog that fills the other 6?
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_probe.hip -o tools/xcd_probe && tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15; }

__global__ void xcc_ids(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// logical block index of this workgroup under an XCD mask (hardware sends workgroup i to XCD i % 8), or -1
__device__ __forceinline__ int masked_block(unsigned mask, int nact) {
  const int x = blockIdx.x & 7;
  if (!((mask >> x) & 1u)) return -1;
  return (blockIdx.x >> 3) * nact + __popc(mask & ((1u << x) - 1u));
}

__global__ __launch_bounds__(1024) void stream_probe(const float* __restrict__ w, float* __restrict__ out, int chunks, int blocks, unsigned mask, int nact) {
  const int lb = masked_block(mask, nact);
  if (lb < 0 || lb >= blocks) return;
  const int wave = (lb * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const float* base = w + (size_t)wave * chunks * 256 + lane * 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int c = 0; c < chunks; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = f32x4{0, 0, 0, 0}; if (c + u < chunks) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(c + u) * 256)); }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 123.456f) out[wave] = s;
}

// compute hog: 512 threads, 96 KiB of LDS (one workgroup per CU), spins for `cycles` of s_memrealtime
__global__ __launch_bounds__(512) void hog(unsigned mask, long long ticks, float* out) {
  extern __shared__ float sm[];
  if (!((mask >> (blockIdx.x & 7)) & 1u)) return;
  const long long t0 = wall_clock64();
  float a = threadIdx.x;
  while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f; }
  sm[threadIdx.x] = a;
  if (a == 0.123f) out[0] = sm[0];
}

static int popc(unsigned m) { return __builtin_popcount(m); }

int main() {
  int* ids; CK(hipMalloc(&ids, 4096 * sizeof(int)));
  hipLaunchKernelGGL(xcc_ids, dim3(64), dim3(64), 0, 0, ids);
  std::vector<int> h(64);
  CK(hipMemcpy(h.data(), ids, 64 * sizeof(int), hipMemcpyDeviceToHost));
  printf("xcc id of workgroups 0..31:");
  bool rr = true;
  for (int i = 0; i < 64; ++i) { if (i < 32) printf(" %d", h[i]); rr = rr && (h[i] == h[i & 7]); }
  printf("\nround-robin with period 8: %s\n", rr ? "yes" : "NO");

  const size_t pool_bytes = (size_t)3 << 30;
  float* pool; float* out;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&out, 64 << 20));
  CK(hipMemset(pool, 0, pool_bytes));
  const size_t pf = pool_bytes / 4;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  const size_t bytes = 26214400, floats = bytes / 4;
  for (int with_hog = 0; with_hog < 2; ++with_hog)
    for (unsigned mask : {0xFFu, 0xC0u, 0xE0u, 0xF0u})
      for (int chunks : {5, 20}) {
        const int nact = popc(mask);
        const int blocks = (int)(floats / ((size_t)chunks * 256) / 16);         // 1024-thread blocks
        const int grid = ((blocks + nact - 1) / nact) * 8;
        const unsigned hog_mask = with_hog ? (~mask & 0xFFu) : 0u;
        if (with_hog && hog_mask == 0) continue;
        const int reps = 200;
        size_t off = 0;
        if (with_hog) hipLaunchKernelGGL(hog, dim3(256), dim3(512), 96 * 1024, sa, hog_mask, (long long)100 * 1000 * 20, out);   // ~20 ms at 100 MHz
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < reps; ++i) {
          hipLaunchKernelGGL(stream_probe, dim3(grid), dim3(1024), 0, sb, pool + off, out, chunks, blocks, mask, nact);
          off = (off + floats) % (pf - floats);
        }
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        const double us = ms * 1e3 / reps;
        printf("stream 26.2MB mask=%02x (%d XCDs) chunks=%2d blocks=%4d grid=%4d %s: %7.2f us/launch %7.1f GB/s\n", mask, nact, chunks, blocks, grid,
               with_hog ? "beside hog on the other XCDs" : "alone                       ", us, bytes / us / 1e3);
      }
  // hog on ALL XCDs beside the stream (the un-partitioned case)
  {
    const int chunks = 5, blocks = (int)(floats / ((size_t)chunks * 256) / 16);
    hipLaunchKernelGGL(hog, dim3(512), dim3(512), 96 * 1024, sa, 0xFFu, (long long)100 * 1000 * 5, out);
    CK(hipEventRecord(e0, sb));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(stream_probe, dim3(blocks), dim3(1024), 0, sb, pool, out, chunks, blocks, 0xFFu, 8);
    CK(hipEventRecord(e1, sb));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipDeviceSynchronize());
    printf("stream 26.2MB unmasked beside a hog on ALL XCDs (2 waves of 256 workgroups x 5 ms): %7.2f us/launch\n", ms * 1e3 / 50);
  }
  return 0;
}
