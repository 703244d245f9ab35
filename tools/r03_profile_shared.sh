set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r03a
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv
MLBP_BENCH_SPINUP_STEPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_tl -- python3 bench.py --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5 --no-cpu-baseline --no-skip-unchanged > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_tl -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_user_k3_trainlayout_nowriteback_b8192.csv
python3 bench.py --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_user_k3_trainlayout_nowriteback_b8192.json 2>/dev/null
python3 tools/time_train_step.py > gpurun_out/${T}_time_train_step.txt 2>&1
rm -rf gpurun_out/${T}_prof_train gpurun_out/${T}_prof_tl
cat gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv | cut -c1-160
cat gpurun_out/${T}_kernel_stats_user_k3_trainlayout_nowriteback_b8192.csv | cut -c1-160
tail -5 gpurun_out/${T}_time_train_step.txt
