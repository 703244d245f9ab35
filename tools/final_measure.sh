set -e
export TMPDIR=/tmp
T=${1:-r02m}
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}.err
MLBP_BENCH_SPINUP_STEPS=300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_user_k3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-skip-unchanged > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_user_k3 -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_user_k3_b8192.csv
tools/pmc_passes.sh $T user_k3 8192 3 > gpurun_out/${T}_pmc_user_k3.log 2>&1
tools/pmc_passes.sh $T chain8 1024 10 > gpurun_out/${T}_pmc_chain8.log 2>&1
tools/pmc_passes.sh $T ring8 1024 10 > gpurun_out/${T}_pmc_ring8.log 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench2_user_k3_b8192.json 2>/dev/null
MLBP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --batch 2048 > gpurun_out/${T}_bench_user_k3_b2048_gloo2_selflaunch.json 2>/dev/null
for cfg in "chain8 1024" "ring8 1024" "chain8 8192" "user_k4 8192" "ring8_x1000 256" "ring8_x512 1024" "ring8_x512_f32 1024" "ring8_x512_shared 8192" "ring8_x512_shared_f32 8192"; do set -- $cfg; python3 bench.py --workload $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_$1_b$2.json 2>/dev/null; done
MLBP_BENCH_SPINUP_STEPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_x1000 -- python3 bench.py --workload ring8_x1000 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-skip-unchanged > /dev/null 2>&1
cp $(find gpurun_out/${T}_prof_x1000 -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_ring8_x1000_b256.csv
for w in "user_k3_shared" "user_k3_trainlayout"; do python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_b8192.json 2>/dev/null; python3 bench.py --workload $w --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_nowriteback_b8192.json 2>/dev/null; done
cp profiles/pmc_traffic.json gpurun_out/${T}_pmc_traffic_registry.json
cp profiles/${T}_pmc_*.json gpurun_out/
echo DONE
