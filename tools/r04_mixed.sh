#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/mixed_tests.log 2>&1; tail -3 gpurun_out/mixed_tests.log
timeout -k 10 300 python3 tools/time_mixed_epoch.py 2>&1 | grep -v amdgpu.ids
MLBP_SHARED_NO_PF=1 MLBP_SHARED_NO_P3=1 timeout -k 10 300 python3 tools/time_mixed_epoch.py 2>&1 | grep -v amdgpu.ids
