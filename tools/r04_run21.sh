#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04af}
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
SPIN=0 MLBP_BENCH_SPINUP_STEPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv; rm -rf gpurun_out/${T}_prof_train
python3 - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv')))[:6]:
    print('  %-90s calls %s avg %.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
