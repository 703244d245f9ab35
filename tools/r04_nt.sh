#!/bin/bash
# A/B of the lean kernel's cache policies: bench the default workload with libmlbp.so swapped for the MLBP_LEAN_NT variants
export TMPDIR=/tmp
T=${1:-r04ah}
cp macaronicusermodeling_amd/libmlbp.so /tmp/libmlbp_orig.so
for v in 0 5 7; do
  if [ $v -eq 0 ]; then cp /tmp/libmlbp_orig.so macaronicusermodeling_amd/libmlbp.so; else cp tools/bin/libmlbp_nt$v.so macaronicusermodeling_amd/libmlbp.so; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-epoch > gpurun_out/${T}_bench_nt$v.json 2> gpurun_out/${T}_bench_nt$v.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_nt$v.json').read().strip().splitlines()[-1])
print('NT variant $v: value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3), 'train_step', d['train_step'] and round(d['train_step']['ms'], 4))
PY
done
cp /tmp/libmlbp_orig.so macaronicusermodeling_amd/libmlbp.so
