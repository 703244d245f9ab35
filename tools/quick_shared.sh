python3 bench.py --workload user_k3_trainlayout --no-writeback --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('trainlayout nowb: launch %.4f ms  train_step %.4f' % (d['roofline']['avg_launch_ms'], d['train_step']['ms']))"
python3 tools/time_train_step.py 2>&1 | sed -n 2,4p
