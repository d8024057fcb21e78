#!/bin/bash
# rocprofv3 evidence of one round, on the MI355X box:  bash tools/profile_round.sh r03 [stats|pmc|all]
#   gpurun_out/<rnd>_pipeline_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the default configs[2] command, sequential steps
#   gpurun_out/<rnd>_pipeline_pmc_{fetch,write}_size.json   separate --pmc passes (FETCH_SIZE / WRITE_SIZE), condensed
#   gpurun_out/traffic_<rnd>.json                   HBM bytes per launch, gfx950 corrections applied (tools/make_traffic.py)
# (--decode-geometry narrow: the geometry the default, pipelined bench run selects, so that the kernels and their average durations are the
# ones of its roofline leg.)
# The counter passes serialise every dispatch they collect (~5 ms each): the acoustic kernels are collected on the full
# configs[2] step, the decode kernels (125 launches per token) on a 48-token run of the same batch -- their bytes per launch do
# not depend on the number of tokens beyond the KV length, which the 48-token run under-states (noted in profiles/README.md).
set -o pipefail
R=${1:-r03}
WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 60; echo "[profile_round] $(date +%T) still running"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
ACOUSTIC='gemm_bf16x3|conv1d_|flash_attn|aa_act|ada_rms|rows_norm|split_planes|gemm_tn|conv_post|cfm_'
DECODE='gemv_fx|gemv_pl|decode_attn|sample_greedy|embed_step'
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  rm -rf /tmp/prof_ks
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-overlap --decode-geometry narrow > $OUT/${R}_prof_ks.log 2>&1 || exit 1
  cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $OUT/${R}_pipeline_kernel_stats.csv
  rm -rf /tmp/prof_ks
  echo "kernel stats done"
  # the PIPELINED run itself (what the driver's line times: three decode lanes beside the acoustic stage; kernel durations include the
  # contention), and the merged-decode variant (three 16-utterance requests decoded as ONE 48-row batch on the plane GEMV)
  rm -rf /tmp/prof_kp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kp -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-exact-mode > $OUT/${R}_prof_kp.log 2>&1 || exit 1
  cp $(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1) $OUT/${R}_pipeline_pipelined_kernel_stats.csv
  rm -rf /tmp/prof_kp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kp -- python3 $ROOT/bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-mode --coalesce 3 --decode-lanes 2 > $OUT/${R}_prof_kc.log 2>&1 || exit 1
  cp $(find /tmp/prof_kp -name "*kernel_stats.csv" | head -1) $OUT/${R}_pipeline_coalesced_kernel_stats.csv
  rm -rf /tmp/prof_kp
  echo "pipelined + coalesced kernel stats done"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    lc=$(echo $C | tr A-Z a-z)
    rm -rf /tmp/prof_a /tmp/prof_d
    rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$ACOUSTIC" --output-format csv -d /tmp/prof_a -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-overlap --decode-geometry narrow > $OUT/${R}_prof_${lc}_a.log 2>&1 || exit 1
    echo "$C acoustic pass done"
    rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$DECODE" --output-format csv -d /tmp/prof_d -- python3 $ROOT/bench.py --steps 1 --warmup 0 --codes 48 --no-cpu-baseline --no-roofline --no-overlap --decode-geometry narrow > $OUT/${R}_prof_${lc}_d.log 2>&1 || exit 1
    echo "$C decode pass done"
    mkdir -p /tmp/prof_m && rm -rf /tmp/prof_m/* && i=0
    for f in $(find /tmp/prof_a /tmp/prof_d -name "*_counter_collection.csv"); do i=$((i+1)); cp $f /tmp/prof_m/${i}_counter_collection.csv; done
    python3 - <<PY
import csv, glob
rows, head = [], None
for p in sorted(glob.glob("/tmp/prof_m/*_counter_collection.csv")):
    with open(p) as f:
        r = csv.reader(f); h = next(r); head = head or h; rows += list(r)
with open("/tmp/prof_m/all_counter_collection.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(head); w.writerows(rows)
import os
for p in glob.glob("/tmp/prof_m/[0-9]*_counter_collection.csv"): os.remove(p)
PY
    python3 $ROOT/tools/summarize_prof.py /tmp/prof_m $OUT/${R}_pipeline_pmc_${lc}.json
    rm -rf /tmp/prof_a /tmp/prof_d /tmp/prof_m
  done
  python3 $ROOT/tools/make_traffic.py $OUT/${R}_pipeline_pmc_fetch_size.json $OUT/${R}_pipeline_pmc_write_size.json $OUT/traffic_${R}.json ${R#r}
fi
[ -f $OUT/${R}_pipeline_kernel_stats.csv ] && head -8 $OUT/${R}_pipeline_kernel_stats.csv | cut -c1-160
exit 0
