#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/idxtts.h"
#include "prof.h"

namespace idxtts {

struct Rec { hipEvent_t a, b; int cat; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;            // launches recorded since enable
static std::vector<hipEvent_t> g_pool;     // recycled events
static double g_flops[PROF_MAX], g_bytes[PROF_MAX], g_ms[PROF_MAX];
static long g_count[PROF_MAX];
static hipEvent_t g_open[PROF_MAX];

static std::string g_names[PROF_MAX];
static int g_nnames = 0;

int prof_register(const char* kernel_name) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_nnames; ++i) if (g_names[i] == kernel_name) return i;
  if (g_nnames >= PROF_MAX) return PROF_MAX - 1;     // (cannot happen with the launch sites of this library: ~60 families)
  g_names[g_nnames] = kernel_name;
  return g_nnames++;
}

const char* prof_name(int cat) { return (cat >= 0 && cat < g_nnames) ? g_names[cat].c_str() : "?"; }
bool prof_enabled() { return g_on; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(int cat, hipStream_t stream, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t a = get_event();
  (void)hipEventRecord(a, stream);
  g_open[cat] = a;
  g_flops[cat] += flops;
  g_bytes[cat] += bytes;
  g_count[cat] += 1;
}

void prof_end(int cat, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, stream);
  g_recs.push_back(Rec{g_open[cat], b, cat});
}

static void drain() {   // resolve recorded event pairs into per-family milliseconds
  for (const Rec& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) g_ms[r.cat] += ms;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
}

__global__ void prof_empty_kernel() {}

}  // namespace idxtts

using namespace idxtts;

extern "C" {

// What an event pair around ONE launch reads for a kernel that does nothing: the fixed cost the per-launch timing adds to
// every launch (event records + dispatch), to be subtracted before families made of thousands of 9-us launches are compared
// with families of a few 300-us launches.
int idxtts_profile_event_overhead(void* stream, int launches, double* avg_ms) {
  if (!avg_ms || launches <= 0) return 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(2 * (size_t)launches);
  for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return 1;
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(prof_empty_kernel, dim3(256), dim3(256), 0, st);
  for (int i = 0; i < launches; ++i) {
    (void)hipEventRecord(ev[2 * i], st);
    hipLaunchKernelGGL(prof_empty_kernel, dim3(256), dim3(256), 0, st);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  double tot = 0.0;
  for (int i = 0; i < launches; ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); tot += ms; }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *avg_ms = tot / launches;
  return 0;
}

int idxtts_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (on) {
    drain();
    for (int c = 0; c < PROF_MAX; ++c) { g_flops[c] = g_bytes[c] = g_ms[c] = 0.0; g_count[c] = 0; }
  }
  g_on = on != 0;
  return 0;
}

int idxtts_profile_num_kernels(void) { std::lock_guard<std::mutex> lk(g_mu); return g_nnames; }

const char* idxtts_profile_kernel_name(int index) { return prof_name(index); }

int idxtts_profile_read(int index, double* total_ms, double* flops, double* bytes, long* launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (index < 0 || index >= g_nnames) return 1;
  drain();
  if (total_ms) *total_ms = g_ms[index];
  if (flops) *flops = g_flops[index];
  if (bytes) *bytes = g_bytes[index];
  if (launches) *launches = g_count[index];
  return 0;
}

}  // extern "C"
