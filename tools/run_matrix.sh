set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
i=0
for flags in "" "--acoustic-workers 2" "--acoustic-workers 3" "--side-by-side" "--side-by-side --acoustic-workers 2"; do
  i=$((i+1))
  timeout -k 10 240 python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-roofline $flags > gpurun_out/my_$i.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/my_$i.log | head -1)
  eq=$(grep -o 'bit for bit: [A-Za-z]*' gpurun_out/my_$i.log | head -1)
  echo "S$i [$flags] rc=$rc $v $eq"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/my_$i.log; break; fi
done
timeout -k 10 200 python -m pytest tests/test_checkpoint_gpu.py -m gpu -x -q 2>&1 | tail -5
