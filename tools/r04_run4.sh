#!/bin/bash
# Round-4 run 4: K4 fused gradient + potentials-side expectations: tests, train-step timings, lean epilogue probe
export TMPDIR=/tmp
T=r04e
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gpu_tests.log
timeout -k 10 200 python3 tools/time_train_step.py > gpurun_out/${T}_time_train_step.txt 2>&1; head -6 gpurun_out/${T}_time_train_step.txt
timeout -k 10 200 python3 tools/time_train_step.py --k4 > gpurun_out/${T}_time_train_step_k4.txt 2>&1; cat gpurun_out/${T}_time_train_step_k4.txt
for w in user_k4_shared user_k3_trainlayout; do
  timeout -k 10 200 python3 bench.py --workload $w --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_nowriteback_b8192.json 2> gpurun_out/${T}_bench_$w.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_${w}_nowriteback_b8192.json').read().strip().splitlines()[-1])
r = d['roofline']
print('$w nowriteback', 'ms/step', round(d['ms_per_step'], 4), 'launch', round(r['avg_launch_ms'], 4), r['unit'], round(r['achieved'], 2), 'frac', round(r['frac'], 3), 'train_step', d['train_step'] and round(d['train_step']['ms'], 4), d['train_step'] and d['train_step']['max_abs_difference_to_standalone_gradient_kernel'])
PY
done
timeout -k 10 300 python3 tools/lean_probe.py user_k3 8192 --gradient > gpurun_out/${T}_lean_probe_gradient.txt 2>&1; tail -14 gpurun_out/${T}_lean_probe_gradient.txt
echo DONE
