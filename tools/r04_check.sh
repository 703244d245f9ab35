#!/bin/bash
# One gpurun call's worth of checks behind a change to the shared-table kernels: the GPU test suite, the trainer step's times
# (K3 / K4), the two train-layout bench lines and the kernel trace of the K3 trainer step.   usage: bash tools/r04_check.sh <tag>
export TMPDIR=/tmp
T=${1:-r04chk}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_tests.log
timeout -k 10 200 python3 tools/time_train_step.py > gpurun_out/${T}_time_train_step.txt 2>&1; head -5 gpurun_out/${T}_time_train_step.txt | tail -4
timeout -k 10 200 python3 tools/time_train_step.py --k4 > gpurun_out/${T}_time_train_step_k4.txt 2>&1; tail -4 gpurun_out/${T}_time_train_step_k4.txt
for w in user_k3_trainlayout user_k4_trainlayout; do
  timeout -k 10 200 python3 bench.py --workload $w --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_nowriteback_b8192.json 2> gpurun_out/${T}_bench_$w.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_${w}_nowriteback_b8192.json').read().strip().splitlines()[-1])
r = d['roofline']
print('$w nowriteback', 'ms/step', round(d['ms_per_step'], 4), 'launch', round(r['avg_launch_ms'], 4), r['unit'], round(r['achieved'], 2), 'frac', round(r['frac'], 3), 'train_step', d['train_step'] and round(d['train_step']['ms'], 4))
PY
done
EXTRA="" bash tools/r04_kt.sh $T user_k3_trainlayout 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare|fused_kernel"
