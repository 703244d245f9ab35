#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04r}
timeout -k 10 900 python3 -m pytest tests/test_gpu_shared.py -m gpu -x -q > gpurun_out/${T}_gpu_tests_shared.log 2>&1
echo "pytest shared rc=$?"; tail -15 gpurun_out/${T}_gpu_tests_shared.log
bash tools/r04_kt.sh $T user_k3_trainlayout 4096 8192
