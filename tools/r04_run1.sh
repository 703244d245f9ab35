#!/bin/bash
# Round-4 run 1: the whole GPU suite on the reworked large-state shared-table path, then the new bench workloads.
export TMPDIR=/tmp
T=r04a
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${T}_gpu_tests.log
tail -5 gpurun_out/${T}_gpu_tests.log
for cfg in "ring8_x512_shared 8192" "ring8_x1000_shared 8192" "ring8_x2048_shared 4096" "ring8_x512_shared_f32 8192"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --workload $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_$1_b$2.json 2> gpurun_out/${T}_bench_$1.err
  python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${T}_bench_$1_b$2.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print('$1 b$2', 'ms/step', round(d['ms_per_step'], 3), 'launch', round(r['avg_launch_ms'], 3), r['unit'], round(r['achieved'], 2), 'frac', round(r['frac'], 3))
except Exception as e:
    print('$1 failed', e); print(open('gpurun_out/${T}_bench_$1.err').read()[-800:])
PY
done
echo DONE
