#!/bin/bash
for S in 1 2 3 6 12; do
  echo "#### sweeps $S"
  EXTRA="--sweeps $S" bash tools/r04_kt.sh r04t user_k3_trainlayout 4096 8192 2>&1 | grep -E "==|sweep_x64_shared|prepare"
done
