set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpt_gpu.py tests/test_serving_gpu.py tests/test_gpt_ref_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/t_kv2.log 2>&1 || { tail -30 gpurun_out/t_kv2.log; exit 1; }
tail -2 gpurun_out/t_kv2.log
timeout -k 10 300 python bench.py --batch 32 --codes 256 --steps 2 --warmup 1 --no-overlap --no-cpu-baseline > gpurun_out/rows2_32.log 2>&1 || { tail -5 gpurun_out/rows2_32.log; exit 1; }
timeout -k 10 300 python bench.py --batch 48 --codes 256 --steps 2 --warmup 1 --no-overlap --no-cpu-baseline > gpurun_out/rows2_48.log 2>&1 || { tail -5 gpurun_out/rows2_48.log; exit 1; }
for f in gpurun_out/rows2_32.log gpurun_out/rows2_48.log; do python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[1], d['value'], d['stage_seconds'], {k:v for k,v in r['kernel_ms_per_step'].items() if 'decode_attn' in k or 'gemv' in k})
PY
done
i=0
for flags in "--coalesce 2 --decode-lanes 2" "--coalesce 2 --decode-lanes 3" "--coalesce 3 --decode-lanes 1" "--coalesce 3 --decode-lanes 2" "--coalesce 2 --decode-lanes 2 --acoustic-workers 2"; do
  i=$((i+1))
  timeout -k 10 400 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline $flags > gpurun_out/co_$i.log 2>&1
  rc=$?
  echo "C$i [$flags] rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/co_$i.log | head -1) $(grep -o 'bit for bit: [A-Za-z]*' gpurun_out/co_$i.log | head -1)"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/co_$i.log; exit 1; fi
done
