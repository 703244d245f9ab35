#!/bin/bash
# HBM traffic of one bench.py workload, measured as MI355X_MICROARCH.md's HBM section prescribes: two SEPARATE
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; with --kernel-trace only), summarised by tools/pmc_traffic.py
# into profiles/<tag>_pmc_fetch_write_<workload>_b<batch>.json and merged into profiles/pmc_traffic.json together
# with the sha of the kernel sources the passes ran on (bench.py reports `traffic` only when that sha is current).
# usage (from the repo root, on the GPU box):  tools/pmc_passes.sh <tag> <workload> <batch> <sweeps> [extra bench args]
set -e
tag=$1; wl=$2; batch=$3; sweeps=$4; shift 4
export TMPDIR=/tmp
export MLBP_BENCH_SPINUP_STEPS=0     # counters are per dispatch: no need for steady clocks, and 300 profiled launches are slow
out=gpurun_out/pmc_${tag}_${wl}_b${batch}
mkdir -p $out profiles
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 bench.py --workload $wl --batch $batch --steps 5 --warmup 1 --no-cpu-baseline --no-skip-unchanged "$@" > $out/$c.log 2>&1
done
f=$(find $out/FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find $out/WRITE_SIZE -name '*counter_collection.csv' | head -1)
key=$wl
for x in "$@"; do [ "$x" = "--no-writeback" ] && key=${wl}_nowriteback; done
python3 tools/pmc_traffic.py $f $w $wl $batch $sweeps profiles/${tag}_pmc_fetch_write_${key}_b${batch}.json ${key}_b${batch}
