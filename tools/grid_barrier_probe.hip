// What would a persistent (one-launch-per-token) decode step pay per phase boundary on MI355X?  Measures a device-wide barrier
// between co-resident workgroups (cooperative launch) and the exchange of a decode-sized activation vector across it:
//   mode 0: barrier only (release add + acquire spin, agent scope)
//   mode 1: barrier + every workgroup writes its 1/G slice of an 80 KiB vector (plain stores, made visible by the release) and
//           reads the WHOLE vector back after the barrier (plain loads after the acquire)
//   mode 2: same exchange with relaxed agent-scope atomic stores / loads for the data and a relaxed counter: no cache maintenance
// Every spin is bounded (abort flag), so a lost workgroup cannot hang the GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Params {
  unsigned* counter;      // monotonic arrival counter
  unsigned* abort_flag;
  float* vec;             // [2][VEC] ping-pong
  float* out;             // [grid] checksums
  int nbar, mode, vec_n;
};

template <bool RELAXED>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* abort_flag, unsigned target) {
  __shared__ int s_ok;
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (RELAXED) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (true) {
      const unsigned v = RELAXED ? __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      if (++spins > (1 << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

__global__ __launch_bounds__(1024) void probe(const Params p) {
  const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x;
  float sum = 0.f;
  const int per = p.vec_n / G;      // slice of this workgroup
  for (int it = 0; it < p.nbar; ++it) {
    float* cur = p.vec + (size_t)(it & 1) * p.vec_n;
    if (p.mode == 1) {
      for (int i = tid; i < per; i += blockDim.x) cur[g * per + i] = (float)(it + 1) + 0.001f * (g * per + i);
    } else if (p.mode == 2) {
      for (int i = tid; i < per; i += blockDim.x)
        __hip_atomic_store(&cur[g * per + i], (float)(it + 1) + 0.001f * (g * per + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned target = (unsigned)(it + 1) * G;
    const bool ok = (p.mode == 2) ? grid_barrier<true>(p.counter, p.abort_flag, target) : grid_barrier<false>(p.counter, p.abort_flag, target);
    if (!ok) return;
    // every element must be THIS iteration's value (a stale line from two iterations ago differs by 2): count mismatches
    if (p.mode == 1) {
      for (int i = tid; i < p.vec_n; i += blockDim.x) sum += (cur[i] != (float)(it + 1) + 0.001f * i) ? 1.f : 0.f;
    } else if (p.mode == 2) {
      for (int i = tid; i < p.vec_n; i += blockDim.x)
        sum += (__hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (float)(it + 1) + 0.001f * i) ? 1.f : 0.f;
    }
  }
  // block checksum
  __shared__ float red[1024];
  red[tid] = sum;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  if (tid == 0) p.out[g] = red[0];
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, probe, 1024, 0));
  printf("CUs %d, co-resident 1024-thread workgroups per CU %d, cooperativeLaunch %d\n", prop.multiProcessorCount, per_cu, prop.cooperativeLaunch);
  const int VEC = 20480;       // 16 rows x 1280 floats = 80 KiB
  unsigned *counter, *abort_flag; float *vec, *out;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&vec, 2 * VEC * 4)); CK(hipMalloc(&out, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {64, 128, 256, 512};
  for (int mode = 0; mode < 3; ++mode)
    for (int gi = 0; gi < 4; ++gi) {
      const int G = grids[gi];
      if (G > per_cu * prop.multiProcessorCount) continue;
      for (int threads : {256, 1024}) {
        Params p{counter, abort_flag, vec, out, 0, mode, VEC};
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          p.nbar = 2000;
          CK(hipMemset(counter, 0, 4)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(vec, 0, 2 * VEC * 4));
          void* args[] = {&p};
          CK(hipEventRecord(e0, 0));
          CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(probe), dim3(G), dim3(threads), args, 0, 0));
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        unsigned ab = 0; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        std::vector<float> h(G); CK(hipMemcpy(h.data(), out, G * 4, hipMemcpyDeviceToHost));
        double stale = 0;
        for (int g = 0; g < G; ++g) stale += h[g];
        printf("mode %d  grid %3d x %4d threads: %7.3f us per barrier%s   abort=%u  stale elements seen %.0f\n", mode, G, threads,
               best * 1e3 / p.nbar, mode ? " (+ 80 KiB exchange)" : "", ab, stale);
      }
    }
  return 0;
}
