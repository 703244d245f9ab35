# Round-3 measurement run (1 x MI355X gpurun box, from the repo root): bench lines, rocprofv3 kernel-trace stats and the
# two separate PMC passes for the workloads quoted in DESIGN.md / profiles/README.md.   usage: bash tools/final_measure_r03.sh <tag> <part>
set -e
export TMPDIR=/tmp
T=${1:-r03f}; PART=${2:-1}
stats() {  # stats <name> <bench args...>: kernel-trace stats csv of `bench.py <args>`
  local name=$1; shift
  MLBP_BENCH_SPINUP_STEPS=${SPIN:-300} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_$name -- python3 bench.py "$@" --no-cpu-baseline --no-skip-unchanged > /dev/null 2>&1
  cp $(find gpurun_out/${T}_prof_$name -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_$name.csv
  rm -rf gpurun_out/${T}_prof_$name
}
if [ "$PART" = 1 ]; then
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}.err
  stats user_k3_b8192 --steps 20 --warmup 5
  tools/pmc_passes.sh $T user_k3 8192 3 > gpurun_out/${T}_pmc_user_k3.log 2>&1
  tools/pmc_passes.sh $T chain8 1024 10 > gpurun_out/${T}_pmc_chain8.log 2>&1
  tools/pmc_passes.sh $T ring8 1024 10 > gpurun_out/${T}_pmc_ring8.log 2>&1
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench2_user_k3_b8192.json 2>/dev/null
  SPIN=0 stats chain8_b1024 --workload chain8 --batch 1024 --steps 20 --warmup 5
  SPIN=0 stats ring8_b1024 --workload ring8 --batch 1024 --steps 20 --warmup 5
  for cfg in "chain8 1024" "ring8 1024" "chain8 8192" "user_k4 8192"; do set -- $cfg; python3 bench.py --workload $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_$1_b$2.json 2>/dev/null; done
  cp profiles/pmc_traffic.json gpurun_out/${T}_pmc_traffic_registry.json
  cp profiles/${T}_pmc_*.json gpurun_out/
else
  for w in user_k3_shared user_k3_trainlayout user_k4_shared; do
    python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_b8192.json 2>/dev/null
    python3 bench.py --workload $w --no-writeback --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_${w}_nowriteback_b8192.json 2>/dev/null
  done
  SPIN=0 stats user_k3_trainlayout_nowriteback_b8192 --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5
  SPIN=0 stats user_k3_shared_b8192 --workload user_k3_shared --steps 20 --warmup 5
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py > /dev/null 2>&1
  cp $(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_train_step_user_k3_b8192.csv
  rm -rf gpurun_out/${T}_prof_train
  python3 tools/time_train_step.py > gpurun_out/${T}_time_train_step.txt 2>&1
  for cfg in "ring8_x512_shared 8192" "ring8_x512_shared_f32 8192" "ring8_x1000 256" "ring8_x512 1024" "ring8_x512_f32 1024"; do set -- $cfg; python3 bench.py --workload $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_$1_b$2.json 2>/dev/null; done
  SPIN=0 stats ring8_x512_shared_b8192 --workload ring8_x512_shared --batch 8192 --steps 5 --warmup 2
  SPIN=0 stats ring8_x512_shared_f32_b8192 --workload ring8_x512_shared_f32 --batch 8192 --steps 5 --warmup 2
  MLBP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --batch 2048 > gpurun_out/${T}_bench_user_k3_b2048_gloo2_selflaunch.json 2>/dev/null
  python3 tools/stamp_shared.py --shared-unary --no-writeback > gpurun_out/${T}_stamps_shared_kernel_trainlayout.txt 2>&1
  python3 tools/stamp_shared.py --gradient > gpurun_out/${T}_stamps_shared_kernel_train_step.txt 2>&1
fi
echo DONE
