// Accuracy of sin^2 through v_sin_f32 (argument in revolutions) against the Cody-Waite + polynomial form of csrc/aa_act.hip, both against double.
//   hipcc -O3 --offload-arch=gfx950 tools/vsin_probe.hip -o /tmp/vsin_probe && /tmp/vsin_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float sin2_poly(float x) {
  const float n = rintf(x * 0.31830988618379067f);
  float r = fmaf(-n, 3.140625f, x);
  r = fmaf(-n, 9.67502593994140625e-4f, r);
  r = fmaf(-n, 1.509957990978376e-7f, r);
  const float r2 = r * r;
  float q = fmaf(r2, -2.5052108385441720e-8f, 2.7557319223985893e-6f);
  q = fmaf(r2, q, -1.9841269841269841e-4f);
  q = fmaf(r2, q, 8.3333333333333332e-3f);
  q = fmaf(r2, q, -1.6666666666666666e-1f);
  const float sn = fmaf(r * r2, q, r);
  return sn * sn;
}
__device__ __forceinline__ float sin2_hw(float x) {
  const float s = __builtin_amdgcn_sinf(x * 0.15915494309189535f);
  return s * s;
}
__device__ __forceinline__ float sin2_hw_fract(float x) {
  const float s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f));
  return s * s;
}
__global__ void k(const float* x, float* a, float* b, float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = sin2_poly(x[i]); b[i] = sin2_hw(x[i]); c[i] = sin2_hw_fract(x[i]); }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), a(n), b(n), c(n);
  for (double range : {1.0, 8.0, 60.0, 400.0}) {
    for (int i = 0; i < n; ++i) x[i] = (float)((2.0 * i / (n - 1) - 1.0) * range);
    float *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
      const double s = std::sin((double)x[i]), r = s * s;
      ea = std::max(ea, std::fabs(a[i] - r)); eb = std::max(eb, std::fabs(b[i] - r)); ec = std::max(ec, std::fabs(c[i] - r));
    }
    printf("|x| <= %6.1f: max abs error of sin^2: Cody-Waite + polynomial %.3e, v_sin_f32 %.3e, v_fract + v_sin_f32 %.3e\n", range, ea, eb, ec);
    hipFree(dx); hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
