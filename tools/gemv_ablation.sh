#!/bin/bash
# What-if measurement on the GPU box's scratch copy (tools/gemv_fx_ablate.py): pipelined configs[2] throughput with the decode GEMV's post-arrival
# arithmetic removed (IDXTTS_FX_DBG=66) against the same build with it (IDXTTS_FX_DBG unset), alternating.  Outputs of the ablated runs are garbage.
#   bash tools/gemv_ablation.sh  ->  gpurun_out/r4/gemv_ablation.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
python3 $ROOT/tools/gemv_fx_ablate.py $ROOT/index-tts_amd/csrc/gemv_fx.hip
(cd $ROOT/index-tts_amd/csrc && make > /dev/null 2>&1)
: > $OUT/gemv_ablation.txt
for v in 0 66 0 66; do
  IDXTTS_FX_DBG=$v python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-mode > $OUT/gemv_ablation_$v.log 2>&1 || true
  tail -1 $OUT/gemv_ablation_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined, IDXTTS_FX_DBG=$v:', d['value'], 'audio-s/s,', d['ms_per_step'], 'ms per step')" >> $OUT/gemv_ablation.txt || echo "run $v failed" >> $OUT/gemv_ablation.txt
done
for v in 0 66; do
  IDXTTS_FX_DBG=$v python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-mode --no-overlap > $OUT/gemv_ablation_seq_$v.log 2>&1 || true
  tail -1 $OUT/gemv_ablation_seq_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sequential, IDXTTS_FX_DBG=$v: gpt_gen', d['stage_seconds']['gpt_gen_time'], 's per batch')" >> $OUT/gemv_ablation.txt || echo "seq run $v failed" >> $OUT/gemv_ablation.txt
done
cat $OUT/gemv_ablation.txt
