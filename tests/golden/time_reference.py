#!/usr/bin/env python3
"""Times the REFERENCE itself (loaded exactly as make_golden.py loads it: in-memory lib2to3 of LBP.py, the .pyx
cythonized in a temp dir) and the oracle port on the benchmark's graph shape, in THIS container only (the reference
never travels to the GPU box).  Cross-check for bench.py's `cpu_baseline` (kind "port"): the port must not be faster
or slower than the real thing by much, or the reported CPU baseline would mislead.

    python tests/golden/time_reference.py [--reference /root/reference] [--graphs 40]

Per graph, as train_mp.py:381-382 does per instance: initialize() + treelike_inference(3) (roots 1, 4, 7 of the K3 user
graph, |X| = 64), graph construction excluded, one process, BLAS threads as configured."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import cases as C  # noqa: E402
from make_golden import load_reference, Roots  # noqa: E402
from oracle import lbp_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reference', default='/root/reference')
ap.add_argument('--graphs', type=int, default=40)
a = ap.parse_args()
L, au, cleanup = load_reference(a.reference)
try:
    spec = C.user_spec(10, [1, 4, 7], 64, 64, seed=1)
    roots = [1, 4, 7]
    graphs = [(C.make_inputs(spec, 1236 + i), None) for i in range(a.graphs)]
    fgs = [C.build_graph(L, spec, inp) for inp, _ in graphs]
    feeder = Roots(L)                           # roots are inputs, as in the fixtures (LBP.py:176, 223 draw them)
    t0 = time.perf_counter()
    for fg in fgs:
        feeder.queue = [roots[0]]               # has_loops() consumes one draw inside initialize()
        fg.initialize()
        fg.isLoopy = True
        feeder.queue = list(roots)
        fg.treelike_inference(3)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    for inp, _ in graphs:
        g = O.Graph(spec)
        msgs = O.init_messages(g)
        for r in roots:
            O.sweep(g, inp, msgs, r)
    t_port = time.perf_counter() - t0
    n = a.graphs * 3
    print('reference (translated in memory): %.1f graph-sweeps/s per core   oracle port: %.1f graph-sweeps/s per core   ratio port/reference %.2f'
          % (n / t_ref, n / t_port, t_ref / t_port))
finally:
    cleanup()
