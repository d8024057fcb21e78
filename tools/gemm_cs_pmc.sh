#!/bin/bash
# Bytes past the L2 of the split-bf16 GEMM under different column-group thresholds (IDXTTS_GEMM_CS_BYTES; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
#   bash tools/gemm_cs_pmc.sh  ->  gpurun_out/r4/gemm_cs_pmc.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/gemm_cs_pmc.txt
for T in 1.5e6 3.2e6 7e6; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_gc
    IDXTTS_GEMM_CS_BYTES=$T rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "gemm_bf16x3_v2_kernel" --output-format csv -d /tmp/prof_gc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-overlap --no-exact-mode > $OUT/gemm_cs_pmc_$T$C.log 2>&1
    f=$(find /tmp/prof_gc -name "*counter_collection.csv" | head -1)
    python3 - "$f" $T $C >> $OUT/gemm_cs_pmc.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[3]]
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][-48:]].append(float(r["Counter_Value"]))
allv = [v for vs in by.values() for v in vs]
print(f"threshold {sys.argv[2]} {sys.argv[3]}: launches {len(allv)}, mean per launch {sum(allv)/len(allv)/1024:.1f} MiB (raw)")
for k, v in sorted(by.items()):
    print(f"    {k}: launches {len(v)}, mean {sum(v)/len(v)/1024:.1f} MiB (raw)")
PY
  done
done
cat $OUT/gemm_cs_pmc.txt
