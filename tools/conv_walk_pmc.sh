#!/bin/bash
# HBM traffic of the wide vocoder convolution under the two tile walks (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes):
#   bash tools/conv_walk_pmc.sh  ->  gpurun_out/r4/conv_walk_pmc.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/conv_walk_pmc.txt
for W in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_cw
    IDXTTS_CONV_WALK=$W rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "conv1d_bf16x3_kernel<2, 2, 2, 2" --output-format csv -d /tmp/prof_cw -- python3 $ROOT/bench.py --workload vocoder --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/conv_walk_pmc_$W$C.log 2>&1
    f=$(find /tmp/prof_cw -name "*counter_collection.csv" | head -1)
    python3 - "$f" $W $C >> $OUT/conv_walk_pmc.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[3]]
v = [float(r["Counter_Value"]) for r in rows]
print(f"walk {sys.argv[2]} {sys.argv[3]}: launches {len(v)}, mean per launch {sum(v)/len(v):.1f} KiB (raw), total {sum(v)/1048576:.2f} GiB (raw)")
print("   per launch, MiB (raw), in launch order: " + " ".join(f"{x/1024:.0f}" for x in v))
PY
  done
done
cat $OUT/conv_walk_pmc.txt
