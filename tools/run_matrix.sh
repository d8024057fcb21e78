set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t_final2.log 2>&1 || { tail -30 gpurun_out/t_final2.log; exit 1; }
tail -2 gpurun_out/t_final2.log
i=0
for flags in "--decode-lanes 2" "--decode-lanes 3" "--decode-lanes 4" "--decode-lanes 4 --lane-priority normal" "--decode-lanes 3 --acoustic-workers 2"; do
  i=$((i+1))
  timeout -k 10 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline $flags > gpurun_out/ln_$i.log 2>&1
  rc=$?
  echo "L$i [$flags] rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/ln_$i.log | head -1) $(grep -o 'bit for bit: [A-Za-z]*' gpurun_out/ln_$i.log | head -1)"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/ln_$i.log; exit 1; fi
done
timeout -k 10 400 python bench.py --workload longform > gpurun_out/r03_longform_bench.log 2>&1 || exit 1
grep -o '"value": [0-9.]*' gpurun_out/r03_longform_bench.log | head -1
