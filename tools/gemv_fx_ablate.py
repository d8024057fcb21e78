"""What-if build for ONE measurement (never shipped; run on the GPU box's scratch copy, tools/gemv_ablation.sh): the decode GEMV with its
post-arrival arithmetic made (almost) free -- no bf16 -> fp32 widening, no LayerNorm row sums, one FMA per chunk instead of four fp32 MFMAs,
plus the 15 bf16 MFMAs a register-direct bf16-plane kernel would issue per wave -- to see how much of the pipelined step is the SIMD issue
time the decode lanes take from the MFMA-bound acoustic stage.  Results of such a build are garbage (finite); only the timing means anything.
IDXTTS_FX_DBG=66 switches it on."""
import re, sys
p = sys.argv[1]
s = open(p).read()
s = s.replace('#include "gemv16.h"', '#define GEMV_PROBE 1\n#include <cstdlib>\n#include "gemv16.h"', 1) if '#define GEMV_PROBE 1' not in s else s
s = s.replace("  p.act = a.act; p.dbg = a.dbg;", '  static const int env_dbg = [] { const char* e = getenv("IDXTTS_FX_DBG"); return e ? atoi(e) : 0; }();\n  p.act = a.act; p.dbg = a.dbg | env_dbg;')
old = """      if (ln && cb + u < nch) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) {"""
new = """      if (ln && cb + u < nch && !FX_DBG(64)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) {"""
assert old in s
s = s.replace(old, new)
old = """      f32x4 wf[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) wf[j] = WRaw<WT>::widen(wq[buf][u][j]);
      if (FX_DBG(2)) {"""
new = """      f32x4 wf[NTW];
      if (FX_DBG(64)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const float c = 1e-3f * (float)(__builtin_bit_cast(unsigned char, (char)(((const char*)&wq[buf][u][j])[0]))); wf[j] = f32x4{c, c, c, c}; }
      } else {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wf[j] = WRaw<WT>::widen(wq[buf][u][j]);
      }
      if (FX_DBG(2)) {"""
assert old in s
s = s.replace(old, new)
open(p, "w").write(s)
print("patched", p)
