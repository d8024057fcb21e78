// How many DEPENDENT kernel dispatches per second does the GPU retire in aggregate when several streams each replay a hipGraph that is
// a linear chain of small kernels -- the shape of N concurrent decode lanes (123 launches per token each)?
//   hipcc -O3 --offload-arch=gfx950 tools/dispatch_rate_probe.hip -o tools/dispatch_rate_probe
// Kernel: `wgs` workgroups of 512 threads that each load `kb` KiB (0 = nothing) and store one word: a stand-in for a decode GEMV's footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void chain_kernel(const f32x4v* __restrict__ src, float* dst, int n16) {
  float acc = 0.f;
  const f32x4v* p = src + (size_t)blockIdx.x * n16 * 512 + threadIdx.x;
  for (int i = 0; i < n16; ++i) { const f32x4v v = __builtin_nontemporal_load(p + (size_t)i * 512); acc += v[0] + v[1] + v[2] + v[3]; }
  if (acc == 123.456f) dst[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int chain = 492, reps = 20;
  f32x4v* src; float* dst;
  const size_t src_bytes = (size_t)1 << 30;
  CK(hipMalloc(&src, src_bytes)); CK(hipMemset(src, 0, src_bytes));
  CK(hipMalloc(&dst, 4096 * 4));
  for (int kb : {0, 40, 128})
    for (int wgs : {256}) {
      const int n16 = kb * 1024 / (512 * 16);      // float4 loads per thread
      for (int nstreams : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(nstreams);
        std::vector<hipGraphExec_t> ex(nstreams);
        for (int s = 0; s < nstreams; ++s) {
          CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
          hipGraph_t g;
          CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
          // every launch of a chain streams its OWN region of the stream's 128 MiB slice (wrapping): like a decode layer's weights, the
          // bytes come from HBM, not from a cache that the previous launch warmed
          const size_t slice = (src_bytes / 16) / 8, per = (size_t)wgs * n16 * 512;
          for (int k = 0; k < chain; ++k)
            hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(512), 0, st[s], src + (size_t)s * slice + (per ? ((size_t)k * per) % (slice - per) : 0), dst, n16);
          CK(hipStreamEndCapture(st[s], &g));
          CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
          CK(hipGraphDestroy(g));
        }
        for (int s = 0; s < nstreams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < reps; ++r)
          for (int s = 0; s < nstreams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
        for (int s = 1; s < nstreams; ++s) { hipEvent_t d; CK(hipEventCreate(&d)); CK(hipEventRecord(d, st[s])); CK(hipStreamWaitEvent(st[0], d, 0)); }
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)chain * reps * nstreams;
        printf("kernel %3d WGs x 512 threads, %2d KiB per WG | %d stream(s): %8.3f us per dependent launch per stream, aggregate %7.1f k launches/s\n", wgs, kb, nstreams,
               ms * 1e3 / (chain * reps), n / ms);
        for (int s = 0; s < nstreams; ++s) { CK(hipGraphExecDestroy(ex[s])); CK(hipStreamDestroy(st[s])); }
      }
    }
  return 0;
}
