#!/bin/bash
# Round-4 run 6: masked minibatches + the bench line's train_epoch object; K4 in the trainer's layout
export TMPDIR=/tmp
T=r04i
timeout -k 10 900 python3 -m pytest tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gpu_tests.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_user_k3_b8192.json').read().strip().splitlines()[-1])
print('user_k3 value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'frac', round(d['roofline']['frac'], 3), 'train_step', round(d['train_step']['ms'], 4))
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('train_epoch', json.dumps(d.get('train_epoch'), indent=1))
PY
tail -5 gpurun_out/${T}_bench.err
for w in user_k4_trainlayout; do
  timeout -k 10 200 python3 bench.py --workload $w --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_nowriteback_b8192.json 2> gpurun_out/${T}_bench_$w.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_${w}_nowriteback_b8192.json').read().strip().splitlines()[-1])
r = d['roofline']
print('$w nowriteback', 'ms/step', round(d['ms_per_step'], 4), 'launch', round(r['avg_launch_ms'], 4), r['unit'], round(r['achieved'], 2), 'frac', round(r['frac'], 3), 'train_step', d['train_step'] and round(d['train_step']['ms'], 4), d['train_step'] and d['train_step']['max_abs_difference_to_standalone_gradient_kernel'])
PY
done
echo DONE
