set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r03b}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv
rm -rf gpurun_out/${T}_prof_train
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv')):
    print('%-70s %5s %10.1f' % (r['Name'].replace('mlbp::(anonymous namespace)::','').replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])))
PY
