#!/bin/bash
# kernel trace of the mixed-shape epoch (tools/time_mixed_epoch.py; extra arguments, e.g. --k1, are passed on)
export TMPDIR=/tmp
T=${1:-r04q}; shift
rm -rf gpurun_out/${T}_prof_mixed
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_mixed -- python3 tools/time_mixed_epoch.py "$@" > gpurun_out/${T}_prof_mixed.log 2>&1
cp $(find gpurun_out/${T}_prof_mixed -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_mixed_epoch.csv
rm -rf gpurun_out/${T}_prof_mixed
python3 - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/${T}_kernel_stats_mixed_epoch.csv')))[:16]:
    print('%-110s calls %5s avg %8.2f us  %5.1f%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
