#!/bin/bash
# K4 (three-source product-fused form) against the general form: kernel traces and the trainer step
export TMPDIR=/tmp
T=${1:-r04p3}
for NO in 0 1; do
  if [ $NO -eq 1 ]; then export MLBP_SHARED_NO_P3=1; fi
  echo "== MLBP_SHARED_NO_P3=$NO"
  EXTRA="" bash tools/r04_kt.sh ${T}_$NO user_k4_trainlayout 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare|ms/step"
  timeout -k 10 200 python3 tools/time_train_step.py --k4 > gpurun_out/${T}_${NO}_time_train_step_k4.txt 2>&1; head -5 gpurun_out/${T}_${NO}_time_train_step_k4.txt | tail -4
done
