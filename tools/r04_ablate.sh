#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04ac}; shift
for M in "$@"; do
  rm -rf gpurun_out/${T}_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_ab -- python3 tools/ablate_shared.py $M $ABARGS > /dev/null 2> gpurun_out/${T}_ab.err
  f=$(find gpurun_out/${T}_ab -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
out = []
for r in rows:
    n = r['Name']
    for key, tag in (('sweep_x64_shared', 'sweep'), ('shared_prepare', 'prepare'), ('sweep_x64_fused', 'fixup')):
        if key in n: out.append('%s %.2f' % (tag, float(r['AverageNs']) / 1e3))
print('mask %5d: ' % $M + '  '.join(out))
PY
  rm -rf gpurun_out/${T}_ab
done
