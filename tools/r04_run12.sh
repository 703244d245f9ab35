#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04n_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04n_gpu_tests.log
bash tools/final_measure_r04.sh r04m 1 > gpurun_out/r04m_part1.log 2>&1; tail -3 gpurun_out/r04m_part1.log
echo DONE
