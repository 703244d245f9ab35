#!/bin/bash
# Diagnostic: one short bench line per workload (launch time, roofline fraction, and the skip-unchanged figure).
for spec in "user_k3 8192" "user_k3_trainlayout 8192" "user_k4 8192" "chain8 1024" "chain8 8192" "ring8 1024"; do
  set -- $spec
  timeout -k 10 200 python3 bench.py --workload $1 --batch $2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('skip_unchanged') or {}
print('$1 b$2', 'step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3),
      '| skip_unchanged launch', k.get('avg_launch_ms'), 'dropped', k.get('updates_dropped'), 'of', k.get('updates_in_schedule'), 'identical', k.get('marginals_bit_identical_to_full_schedule'), k.get('max_abs_marginal_difference'))"
done
