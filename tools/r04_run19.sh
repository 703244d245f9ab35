#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04x}
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_gpu_tests.log
SWEEPS="1 3" BATCHES="8192" bash tools/r04_run17.sh $T
