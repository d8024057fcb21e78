#!/bin/bash
# Instruction-mix counters of the pipeline's top kernels (rocprofv3 --pmc, one pass per counter pair) on a short sequential configs[2] step:
#   bash tools/top_kernels_pmc.sh  ->  gpurun_out/r4/top_pmc.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/top_pmc.txt
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM"; do
  rm -rf /tmp/prof_tk
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "gemv_fx_kernel|decode_attn16_kernel|flash_attn_planes_kernel|gemm_bf16x3_v2_kernel<false, 0|conv1d_bf16x3_kernel<1, 4|sample_greedy" --output-format csv -d /tmp/prof_tk -- python3 $ROOT/bench.py --steps 1 --warmup 0 --codes 64 --no-cpu-baseline --no-roofline --no-overlap --no-exact-mode --decode-geometry narrow > $OUT/top_pmc_run.log 2>&1 || { echo "counter set $C failed" >> $OUT/top_pmc.txt; continue; }
  f=$(find /tmp/prof_tk -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/top_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][-64:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    for c, v in cs.items():
        print(f"{name} | {c}: launches {len(v)}, mean {sum(v)/len(v):.4g}")
PY
done
cat $OUT/top_pmc.txt
