#!/bin/bash
# SQ / LDS counters of the sweep kernels of one bench.py workload (diagnostic; two --pmc passes, 8 SQ slots each).
# usage (repo root, GPU box): tools/sq_counters.sh <tag> [bench args]
set -e
tag=$1; shift
export TMPDIR=/tmp
export MLBP_BENCH_SPINUP_STEPS=0
out=gpurun_out/sq_${tag}
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $out/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'sweep' not in k and 'gemm' not in k and 'contract' not in k:
            continue
        acc[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-24s %14.0f  (mean of %d dispatches)' % (c, sum(v) / len(v), len(v)))
PY
