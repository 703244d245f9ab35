#!/bin/bash
T=${1:-r04t}
for S in ${SWEEPS:-1 3 6}; do
  echo "#### sweeps $S"
  EXTRA="--sweeps $S" bash tools/r04_kt.sh $T user_k3_trainlayout ${BATCHES:-4096 8192} 2>&1 | grep -E "==|sweep_x64_shared|prepare|fused_kernel"
done
