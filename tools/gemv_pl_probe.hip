// Cold-weight timing of the plane GEMV (csrc/gemv_pl.hip) on the GPT's decode shapes, by row count and workgroup geometry.
// Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -DPL_STAMPS -Iindex-tts_amd/csrc tools/gemv_pl_probe.hip index-tts_amd/csrc/gemv_pl.hip index-tts_amd/csrc/gemv_fx.hip \
//         index-tts_amd/csrc/prof.hip -o /tmp/gemv_pl_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemv_pl.h"
#include <algorithm>
using namespace idxtts;
namespace idxtts { void gemv_pl_set_stamps(unsigned long long* buf); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
namespace idxtts { int fail(const char* file, int line, const std::string& msg) { printf("%s:%d %s\n", file, line, msg.c_str()); return 1; } }

struct Shape { const char* tag; int N, K; bool ln, res, planes, rowmajor; };

int main(int argc, char** argv) {
  const size_t pool_bytes = (size_t)1 << 30;
  char* pool; unsigned short* xp; float *y, *g, *stats, *slab; unsigned* cnt; unsigned short* yp;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
  CK(hipMalloc(&xp, (size_t)64 * 5120 * 4)); CK(hipMemset(xp, 0, (size_t)64 * 5120 * 4));
  CK(hipMalloc(&yp, (size_t)64 * 5120 * 4));
  CK(hipMalloc(&y, 64 * 8448 * 4)); CK(hipMemset(y, 0, 64 * 8448 * 4));
  CK(hipMalloc(&g, 8448 * 4)); CK(hipMemset(g, 0, 8448 * 4));
  CK(hipMalloc(&stats, 320 * 64 * 2 * 4)); CK(hipMemset(stats, 0, 320 * 64 * 2 * 4));
  CK(hipMalloc(&slab, (size_t)64 << 20));
  CK(hipMalloc(&cnt, 1024 * 4)); CK(hipMemset(cnt, 0, 1024 * 4));
  const Shape shapes[] = {{"c_attn", 3840, 1280, true, false, false, true}, {"c_proj", 1280, 1280, false, true, true, true},
                          {"c_fc", 5120, 1280, true, false, true, false}, {"fc2", 1280, 5120, false, true, true, true},
                          {"head", 8194, 1280, false, false, false, true}};
  unsigned long long* stamps = nullptr;
  CK(hipMalloc(&stamps, 4096 * 16 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rows : {16, 32, 48, 64})
    for (const Shape& sh : shapes)
      for (int ct : {1, 2, 4, 8}) {
        gemv_pl_set_ct_override(ct);
        int ctp, kp;
        gemv_pl_plan(sh.N, sh.K, rows, &ctp, &kp);
        if (ctp != ct) continue;      // not a valid geometry for this shape
        const size_t wbytes = gemv32_packed_elems(sh.N, sh.K) * 2;
        size_t off = 0; const int reps = 40;
        for (int it = -3; it < reps; ++it) {
          if (it == 0) CK(hipEventRecord(e0, 0));
          Gemv32Weights w; w.wp = pool + off; w.N = sh.N; w.K = sh.K; w.fmt = WFMT_BF16;
          off += (wbytes + 255) & ~(size_t)255; if (off + wbytes > pool_bytes) off = 0;
          GemvPLArgs a; a.x = reinterpret_cast<float*>(xp); a.ldx = sh.K; a.rows = rows; a.bias = g; a.slab = slab; a.counters = cnt;
          if (sh.ln) { a.colsum = g; a.stats_in = stats; a.stats_tiles = sh.K / 16; }
          a.y = y; a.ldy = sh.N;
          if (sh.res) a.res = y;
          if (sh.planes && sh.res) a.stats_out = stats;
          if (sh.tag[2] == 'f' && sh.tag[1] == '_') a.act = 1;
          if (gemv_pl_forward(w, a, 0)) { printf("gemv failed\n"); return 1; }
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const int wgs = ((sh.N + 15) / 16 + ct - 1) / ct * kp;
        printf("rows %2d %-7s N=%5d K=%5d ct=%d kparts=%2d wgs=%4d : %7.2f us  %7.1f GB/s(weights)\n", rows, sh.tag, sh.N, sh.K, ct, kp, wgs, us, wbytes / us / 1e3);
        if (stamps && (rows == 16 || rows == 48)) {
          // one more launch with in-kernel stamps: per workgroup [0] realtime at entry, [1..8] s_memtime at the phase boundaries,
          // [9] realtime at the arrival decision, [10] s_memtime / [11] realtime at the very end (last arrivers only)
          CK(hipMemset(stamps, 0, 4096 * 16 * 8));
          gemv_pl_set_stamps(stamps);
          Gemv32Weights w; w.wp = pool + off; w.N = sh.N; w.K = sh.K; w.fmt = WFMT_BF16;
          GemvPLArgs a; a.x = reinterpret_cast<float*>(xp); a.ldx = sh.K; a.rows = rows; a.bias = g; a.slab = slab; a.counters = cnt; a.y = y; a.ldy = sh.N;
          if (sh.ln) { a.colsum = g; a.stats_in = stats; a.stats_tiles = sh.K / 16; }
          if (sh.res) a.res = y;
          if (sh.planes && sh.res) a.stats_out = stats;
          if (gemv_pl_forward(w, a, 0)) return 1;
          CK(hipDeviceSynchronize());
          gemv_pl_set_stamps(nullptr);
          std::vector<unsigned long long> h((size_t)wgs * 16);
          CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull, tend = 0;
          for (int i = 0; i < wgs; ++i) { t0 = std::min(t0, h[i * 16]); tend = std::max(tend, std::max(h[i * 16 + 9], h[i * 16 + 11])); }
          auto med = [&](int a, int b, bool last_only) {
            std::vector<double> v;
            for (int i = 0; i < wgs; ++i) if (h[i * 16 + b] && h[i * 16 + a] && (!last_only || h[i * 16 + 10])) v.push_back((double)(h[i * 16 + b] - h[i * 16 + a]));
            if (v.empty()) return -1.0;
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
          };
          std::vector<double> st;
          for (int i = 0; i < wgs; ++i) st.push_back((h[i * 16] - t0) * 0.01);
          std::sort(st.begin(), st.end());
          printf("      stamps: entry spread med %.2f max %.2f us; first entry -> last exit %.2f us | cycles (median): issue loads %.0f, split+stats %.0f, barrier %.0f, "
                 "mfma %.0f, reduce %.0f, slab store+ack %.0f, arrive %.0f, merge+epilogue (last) %.0f\n", st[st.size() / 2], st.back(), (tend - t0) * 0.01,
                 med(1, 2, false), med(2, 3, false), med(3, 4, false), med(4, 5, false), med(5, 6, false), med(6, 7, false), med(7, 8, false), med(8, 10, true));
        }
      }
  return 0;
}
