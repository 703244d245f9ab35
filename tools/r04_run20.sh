#!/bin/bash
export TMPDIR=/tmp
T=${1:-r04aa}
timeout -k 10 280 python tools/stamp_shared.py 8192 --shared-unary --no-writeback > gpurun_out/${T}_stamps_pf_8192.txt 2>&1; tail -11 gpurun_out/${T}_stamps_pf_8192.txt | head -10
timeout -k 10 280 python tools/stamp_shared.py 4096 --shared-unary --no-writeback > gpurun_out/${T}_stamps_pf_4096.txt 2>&1; tail -11 gpurun_out/${T}_stamps_pf_4096.txt | head -10
SWEEPS="3" BATCHES="8192" bash tools/r04_run17.sh $T
