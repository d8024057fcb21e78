// ablation of the narrow split-bf16 convolution's fixed cost.  hipcc -O3 --offload-arch=gfx950 -Iindex-tts_amd/csrc [-DDBG_*] tools/conv_probe2.hip index-tts_amd/csrc/{conv1d_bf16x3,prof}.hip
#include <cstdio>
#include <vector>
#include "conv1d.h"
using namespace idxtts;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
namespace idxtts { int fail(const char* file, int line, const std::string& msg) { printf("%s:%d %s\n", file, line, msg.c_str()); return 1; } }
int main() {
  const int B = 16;
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int C = cfg ? 96 : 24, T = cfg ? 56320 : 225280, taps = 3;
    ConvWeights cw; cw.M = C; cw.Cin = C; cw.K = taps; cw.nchunk = (C + 15) / 16; cw.ups = 1;
    const size_t wbytes = (size_t)((C + 31) / 32) * cw.nchunk * taps * 2048;
    void* cwp; CK(hipMalloc(&cwp, wbytes)); CK(hipMemset(cwp, 0x3c, wbytes)); cw.wp16 = cwp;
    float *x, *y, *r; const size_t n = (size_t)B * C * T;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&r, n * 4)); CK(hipMemset(x, 0, n * 4)); CK(hipMemset(r, 0, n * 4));
    ConvArgs a; a.x = x; a.y = y; a.res = r; a.B = B; a.T = T; a.dil = 1; a.pad_left = 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (conv1d_bf16x3_forward(cw, a, 0)) return 1;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) if (conv1d_bf16x3_forward(cw, a, 0)) return 1;
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("C=%d T=%d k=%d: %.1f us\n", C, T, taps, ms * 1000 / 20);
  }
  return 0;
}
