#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_shared.py tests/test_gpu_gradient.py -m gpu -x -q > gpurun_out/p3c_tests.log 2>&1; tail -3 gpurun_out/p3c_tests.log
timeout -k 10 200 python3 tools/time_train_step.py --k4 2>&1 | grep "shared pots"
timeout -k 10 200 python3 tools/time_train_step.py 2>&1 | grep "shared pots"
bash tools/r04_p3b.sh r04p3c | head -4
