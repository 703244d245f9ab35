#!/usr/bin/env python3
"""Diagnostic (not product): builds a -DMLBP_ABLATE copy of libmlbp.so under gpurun_out/ and times
the fused sweep launch with one phase removed at a time (interleaved rounds in one process).  The
ablated launches compute wrong results; only the time deltas are read."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'ablate')
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, 'libmlbp_ablate.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math',
                       '-DMLBP_ABLATE', '-shared', '-x', 'hip'] +
                      [os.path.join(csrc, f) for f in ('mlbp_host.cpp', 'mlbp_sweep.hip', 'mlbp_lean.hip', 'mlbp_shared.hip', 'mlbp_gemm.hip', 'mlbp_prims.hip', 'mlbp_grad.hip')] +
                      ['-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_set_ablate_mask.argtypes = [C.c_int]
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'user_k3'
spec, roots, sweeps, seed = bench.workload_spec(workload)
X, B = spec['X'], 8192
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev) + 0.01)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
masks = [(0, 'full kernel'), (1, '- variable product'), (2, '- contraction (partials)'), (4, '- gather of partial sums'),
         (8, '- wave max / rescale key'), (15, '- all four (barrier kept)'), (16, '- barrier only (wrong results)'),
         (1 << 9, '- hoisted unary normalisation'), (1 << 10, '- constant products'), (1 << 11, '- final normalisation pass'),
         (1 << 12, '- marginal read-out'), (1 << 13, '- message write-back'), (15 | (31 << 9), '- all of the above')]
times = {m: [] for m, _ in masks}
for rnd in range(6):
    for m, _ in masks:
        ffi.check(ffi.lib.mlbp_debug_set_ablate_mask(m))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fb.sweep(roots, init=True, marginals=marg); e.record(); torch.cuda.synchronize()
        if rnd:
            times[m].append(s.elapsed_time(e))
if len(sys.argv) > 2:      # sweep-count scan of selected masks
    for m in (0, 15):
        ffi.check(ffi.lib.mlbp_debug_set_ablate_mask(m))
        for ns in (3, 12):
            rr = [roots[i % len(roots)] for i in range(ns)]
            ts = []
            for rnd in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fb.sweep(rr, init=True); e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            print('mask %2d sweeps %2d: %.4f ms' % (m, ns, sorted(ts)[2]))
base = sorted(times[0])[len(times[0]) // 2]
for m, name in masks:
    t = sorted(times[m])[len(times[m]) // 2]
    print('%-46s %.4f ms   delta %+.4f' % (name, t, t - base))
