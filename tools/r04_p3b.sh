#!/bin/bash
# kernel trace of the K4 trainer step
export TMPDIR=/tmp
T=${1:-r04p3}
rm -rf gpurun_out/${T}_prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py --k4 > /dev/null 2>&1
f=$(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${T}_kernel_stats_train_step_user_k4_b8192.csv
rm -rf gpurun_out/${T}_prof_train
cut -d, -f1-4 gpurun_out/${T}_kernel_stats_train_step_user_k4_b8192.csv | cut -c1-160 | head -9
