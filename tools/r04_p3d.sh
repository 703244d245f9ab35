#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/p3d_tests1.log 2>&1; tail -2 gpurun_out/p3d_tests1.log
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/p3d_tests2.log 2>&1; tail -2 gpurun_out/p3d_tests2.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/p3d_tests3.log 2>&1; tail -2 gpurun_out/p3d_tests3.log
bash tools/r04_p3b.sh r04p3d | head -4
