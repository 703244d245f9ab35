#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04v}
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_gpu_tests.log
