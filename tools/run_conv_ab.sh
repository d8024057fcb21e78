set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_vocoder_gpu.py tests/test_pipeline_gpu.py tests/test_serving_gpu.py -m gpu -x -q > gpurun_out/t_conv4.log 2>&1 || { tail -40 gpurun_out/t_conv4.log; exit 1; }
tail -1 gpurun_out/t_conv4.log
timeout -k 10 300 python bench.py --workload vocoder --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/conv_voc2.log 2>&1 || { tail -5 gpurun_out/conv_voc2.log; exit 1; }
python - <<'PY'
import json,sys
for l in open('gpurun_out/conv_voc2.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('voc', d['value'], {k[21:]:(v['achieved'],v['time_share']) for k,v in r['top_kernels'].items() if 'conv' in k})
PY
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/conv_pipe.log 2>&1 || { tail -5 gpurun_out/conv_pipe.log; exit 1; }
python - <<'PY'
import json,sys
for l in open('gpurun_out/conv_pipe.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pipe', d['value'], d['stage_seconds'], {k:r.get(k) for k in ('kernel','frac')}, d['config'].get('outputs_equal_sequential'))
PY
