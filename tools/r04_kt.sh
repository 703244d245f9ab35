#!/bin/bash
# kernel-trace of the shared-table launch sequence at several batch sizes (production build)
export TMPDIR=/tmp
T=${1:-r04q}
W=${2:-user_k3_trainlayout}
shift 2
for B in "$@"; do
  rm -rf gpurun_out/${T}_kt
  MLBP_BENCH_SPINUP_STEPS=50 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_kt -- python3 bench.py --workload $W --no-writeback --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-skip-unchanged --no-train-epoch $EXTRA > gpurun_out/${T}_kt_${W}_b$B.json 2> gpurun_out/${T}_kt.err
  f=$(find gpurun_out/${T}_kt -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/${T}_kernel_stats_${W}_nowriteback_b$B.csv
  echo "== $W B=$B"
  python3 - <<PY
import csv, json
for r in list(csv.DictReader(open('gpurun_out/${T}_kernel_stats_${W}_nowriteback_b$B.csv')))[:5]:
    print('  %-100s calls %s avg %.2f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
try:
    d = json.loads(open('gpurun_out/${T}_kt_${W}_b$B.json').read().strip().splitlines()[-1])
    print('  ms/step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4))
except Exception as e:
    print('  bench line unreadable', e)
PY
  rm -rf gpurun_out/${T}_kt
done
