#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04ak}
timeout -k 10 600 python3 -m pytest tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_tests.log
SWEEPS="3" BATCHES="8192" bash tools/r04_run17.sh $T
MLBP_SHARED_NO_STAGED_PREPARE=1 SWEEPS="3" BATCHES="8192" bash tools/r04_run17.sh ${T}_unstaged
