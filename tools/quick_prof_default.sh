# Diagnostic: kernel-trace stats of the default bench (top kernels).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MLBP_BENCH_SPINUP_STEPS=300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qprof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-skip-unchanged > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/qprof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print('%-80s %6s %10.1f' % (r['Name'].replace('mlbp::(anonymous namespace)::','').replace('(anonymous namespace)::','')[:80], r['Calls'], float(r['AverageNs'])))
PY
rm -rf gpurun_out/qprof
