#!/bin/bash
export TMPDIR=/tmp
T=r04m0
timeout -k 10 900 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gpu_tests.log
for i in 1 2; do
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-epoch > gpurun_out/${T}_bench${i}_user_k3_b8192.json 2> gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench${i}_user_k3_b8192.json').read().strip().splitlines()[-1])
print('user_k3 value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3), 'train_step', round(d['train_step']['ms'], 4))
PY
done
timeout -k 10 200 python3 bench.py --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_user_k3_trainlayout_nowriteback_b8192.json 2>> gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_user_k3_trainlayout_nowriteback_b8192.json').read().strip().splitlines()[-1])
print('trainlayout value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3), 'train_step', round(d['train_step']['ms'], 4))
PY
echo DONE
