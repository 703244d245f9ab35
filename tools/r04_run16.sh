#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04s}
timeout -k 10 900 python3 -m pytest tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/${T}_gpu_tests_shared.log 2>&1
echo "pytest shared rc=$?"; tail -15 gpurun_out/${T}_gpu_tests_shared.log
timeout -k 10 280 python tools/stamp_shared.py 8192 --shared-unary --no-writeback > gpurun_out/${T}_stamps_pf_8192.txt 2>&1; cat gpurun_out/${T}_stamps_pf_8192.txt | tail -12
timeout -k 10 280 python tools/stamp_shared.py 4096 --shared-unary --no-writeback > gpurun_out/${T}_stamps_pf_4096.txt 2>&1; cat gpurun_out/${T}_stamps_pf_4096.txt | tail -12
