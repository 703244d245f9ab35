# Round-4 measurement run (1 x MI355X gpurun box, from the repo root): bench lines, rocprofv3 kernel-trace stats and the two
# separate PMC passes for the workloads quoted in DESIGN.md / profiles/README.md.   usage: bash tools/final_measure_r04.sh <tag> <part>
set -e
export TMPDIR=/tmp
T=${1:-r04q}; PART=${2:-1}
stats() {  # stats <name> <bench args...>: kernel-trace stats csv of `bench.py <args>`
  local name=$1; shift
  MLBP_BENCH_SPINUP_STEPS=${SPIN:-300} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_$name -- python3 bench.py "$@" --no-cpu-baseline --no-skip-unchanged --no-train-epoch > /dev/null 2>&1
  cp $(find gpurun_out/${T}_prof_$name -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_$name.csv
  rm -rf gpurun_out/${T}_prof_$name
}
line() {  # line <workload> <batch> [extra args]: one bench line
  local w=$1 b=$2; shift 2
  local tag=${w}; for x in "$@"; do [ "$x" = "--no-writeback" ] && tag=${w}_nowriteback; done
  python3 bench.py --workload $w --batch $b "$@" --no-cpu-baseline --no-train-epoch > gpurun_out/${T}_bench_${tag}_b$b.json 2>/dev/null
}
if [ "$PART" = 1 ]; then
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_user_k3_b8192.json 2> gpurun_out/${T}.err
  echo "bench done"
  stats user_k3_b8192 --steps 20 --warmup 5
  tools/pmc_passes.sh $T user_k3 8192 3 --no-train-epoch > gpurun_out/${T}_pmc_user_k3.log 2>&1
  tools/pmc_passes.sh $T chain8 1024 10 --no-train-epoch > gpurun_out/${T}_pmc_chain8.log 2>&1
  tools/pmc_passes.sh $T ring8 1024 10 --no-train-epoch > gpurun_out/${T}_pmc_ring8.log 2>&1
  echo "pmc done"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench2_user_k3_b8192.json 2>/dev/null
  SPIN=0 stats chain8_b1024 --workload chain8 --batch 1024 --steps 20 --warmup 5
  SPIN=0 stats ring8_b1024 --workload ring8 --batch 1024 --steps 20 --warmup 5
  for cfg in "chain8 1024" "ring8 1024" "chain8 8192" "user_k4 8192"; do set -- $cfg; line $1 $2 --steps 20 --warmup 5; done
  cp profiles/pmc_traffic.json gpurun_out/${T}_pmc_traffic_registry.json
  cp profiles/${T}_pmc_*.json gpurun_out/
else
  for w in user_k3_shared user_k3_trainlayout user_k4_shared user_k4_trainlayout; do
    line $w 8192 --steps 20 --warmup 5
    line $w 8192 --no-writeback --steps 20 --warmup 5
  done
  echo "shared lines done"
  SPIN=0 stats user_k3_trainlayout_nowriteback_b8192 --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5
  SPIN=0 stats user_k4_trainlayout_nowriteback_b8192 --workload user_k4_trainlayout --no-writeback --steps 20 --warmup 5
  for k in "" "--k4"; do
    n=train_step_user_k3_b8192; [ -n "$k" ] && n=train_step_user_k4_b8192
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_train -- python3 tools/prof_train_step.py $k > /dev/null 2>&1
    cp $(find gpurun_out/${T}_prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats_$n.csv
    rm -rf gpurun_out/${T}_prof_train
  done
  python3 tools/time_train_step.py > gpurun_out/${T}_time_train_step.txt 2>&1
  python3 tools/time_train_step.py --k4 > gpurun_out/${T}_time_train_step_k4.txt 2>&1
  echo "train step done"
  for cfg in "ring8_x512_shared 8192" "ring8_x512_shared_f32 8192" "ring8_x1000_shared 8192" "ring8_x2048_shared 4096" "ring8_x1000 256" "ring8_x512 1024" "ring8_x512_f32 1024"; do set -- $cfg; line $1 $2 --steps 5 --warmup 2; done
  SPIN=0 stats ring8_x512_shared_b8192 --workload ring8_x512_shared --batch 8192 --steps 5 --warmup 2
  SPIN=0 stats ring8_x512_shared_f32_b8192 --workload ring8_x512_shared_f32 --batch 8192 --steps 5 --warmup 2
  SPIN=0 stats ring8_x1000_shared_b8192 --workload ring8_x1000_shared --batch 8192 --steps 3 --warmup 1
  SPIN=0 stats ring8_x2048_shared_b4096 --workload ring8_x2048_shared --batch 4096 --steps 3 --warmup 1
  MLBP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --batch 2048 --no-train-epoch > gpurun_out/${T}_bench_user_k3_b2048_gloo2_selflaunch.json 2>/dev/null
fi
echo DONE
