#!/bin/bash
# Round-end evidence on the MI355X box (after tools/profile_round.sh): rocprofv3 kernel stats of the prompt and infer_default workloads,
# then the default bench line with the fresh traffic file in place.   bash tools/final_profiles.sh r03
set -o pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for W in prompt infer_default; do
  rm -rf /tmp/prof_w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -- python3 $ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${R}_prof_${W}.log 2>&1 || exit 1
  cp $(find /tmp/prof_w -name "*kernel_stats.csv" | head -1) $OUT/${R}_${W}_kernel_stats.csv
  echo "$W kernel stats done"
done
rm -rf /tmp/prof_w
[ -f $OUT/traffic_${R}.json ] && cp $OUT/traffic_${R}.json $ROOT/profiles/traffic_${R}.json
cd $ROOT && python3 bench.py > $OUT/${R}_pipeline_bench.log 2>&1 || exit 1
tail -1 $OUT/${R}_pipeline_bench.log | cut -c1-300
