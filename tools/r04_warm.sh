#!/bin/bash
# warmed-up kernel traces of the train-layout launch sequences (thousands of launches), the general form of K4 beside its
# three-source form, and the mixed-shape epoch
export TMPDIR=/tmp
T=${1:-r04q}
EXTRA="" bash tools/r04_kt.sh ${T}w user_k3_trainlayout 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare|fused_kernel|ms/step"
EXTRA="" bash tools/r04_kt.sh ${T}w user_k4_trainlayout 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare|fused_kernel|ms/step"
MLBP_SHARED_NO_P3=1 EXTRA="" bash tools/r04_kt.sh ${T}g user_k4_trainlayout 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare|fused_kernel|ms/step"
timeout -k 10 300 python3 tools/time_mixed_epoch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_time_mixed_epoch.txt
MLBP_SHARED_NO_PF=1 MLBP_SHARED_NO_P3=1 timeout -k 10 300 python3 tools/time_mixed_epoch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_time_mixed_epoch_general_form.txt
cat gpurun_out/${T}_time_mixed_epoch.txt gpurun_out/${T}_time_mixed_epoch_general_form.txt
