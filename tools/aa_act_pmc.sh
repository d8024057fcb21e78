#!/bin/bash
# VALU / wave counters of the vocoder's activation and narrow convolution kernels (rocprofv3 --pmc, one pass per counter set):
#   bash tools/aa_act_pmc.sh   ->  gpurun_out/r4/aa_pmc_*.csv (per-dispatch counter rows)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_')
  rm -rf /tmp/prof_aa
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "aa_act_kernel|conv1d_bf16x3_kernel<1, 4|conv1d_bf16x3_kernel<2, 2, 1" --output-format csv -d /tmp/prof_aa -- python3 $ROOT/bench.py --workload vocoder --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/aa_pmc_$tag.log 2>&1
  f=$(find /tmp/prof_aa -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$OUT/aa_pmc_$tag.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-70:]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for name, cs in agg.items():
        for c, v in cs.items():
            f.write(f"{name} | {c}: launches {len(v)}, mean {sum(v)/len(v):.4g}, max {max(v):.4g}\n")
print(open(sys.argv[2]).read())
PY
done
