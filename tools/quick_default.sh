# Diagnostic: the default bench line in short (value, ms/step, launch, frac, windows).
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.0f  ms/step %.4f  launch %.4f  frac %.3f  windows %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], [round(w, 4) for w in d['windows_ms_per_step']]))"
