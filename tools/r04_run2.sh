#!/bin/bash
# Round-4 run 2: the lean kernel's new gradient epilogue -- gradient tests, then the default bench line (train_step)
export TMPDIR=/tmp
T=r04c
timeout -k 10 600 python3 -m pytest tests/test_gpu_gradient.py tests/test_gpu_sweep.py -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_user_k3_b8192.json').read().strip().splitlines()[-1])
print('user_k3 value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3))
print('train_step', d['train_step'])
PY
echo DONE
