#!/bin/bash
export TMPDIR=/tmp
T=r04j
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gpu_tests.log
echo DONE
