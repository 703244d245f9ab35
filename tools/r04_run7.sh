#!/bin/bash
# Round-4 run 7: where a config-2 launch spends its time (lean kernel, 6+1 and 8 tables), stamps of the shared-table kernel
export TMPDIR=/tmp
T=r04g
timeout -k 10 400 python3 tools/lean_probe.py chain8 1024 > gpurun_out/${T}_lean_probe_chain8_b1024.txt 2>&1; tail -24 gpurun_out/${T}_lean_probe_chain8_b1024.txt
timeout -k 10 400 python3 tools/lean_probe.py ring8 1024 > gpurun_out/${T}_lean_probe_ring8_b1024.txt 2>&1; tail -12 gpurun_out/${T}_lean_probe_ring8_b1024.txt
timeout -k 10 300 python3 tools/stamp_shared.py --shared-unary --no-writeback > gpurun_out/${T}_stamps_shared_kernel_trainlayout.txt 2>&1; tail -30 gpurun_out/${T}_stamps_shared_kernel_trainlayout.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_user_k3_b8192.json').read().strip().splitlines()[-1])
print('train_epoch', json.dumps(d.get('train_epoch'), indent=1))
PY
echo DONE
