#!/usr/bin/env python3
"""Diagnostic (not product): builds a -DMLBP_LEAN_PROBE copy of libmlbp.so under gpurun_out/ and (i) times the lean
sweep launch with one phase removed at a time (interleaved rounds in one process; ablated launches compute wrong
results, only the deltas are read), (ii) prints where wave 0 of every 64th workgroup spends its shader cycles.
Shares, not lengths: the stamped build serialises what the shipped kernel overlaps."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'lean_probe')
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, 'libmlbp_probe.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
from macaronicusermodeling_amd import build as B_  # noqa: E402
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math',
                       '-DMLBP_LEAN_PROBE', '-shared', '-x', 'hip'] + [os.path.join(csrc, f) for f in B_.SOURCES] + ['-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_lean_probe.argtypes = [C.c_int, C.c_void_p]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

GRAD = '--gradient' in sys.argv          # the sweep call carries the gradient (lean kernel's GRAD instance): stamp 9 then closes the epilogue
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
workload = argv[0] if argv else 'user_k3'
spec, roots, sweeps, seed = bench.workload_spec(workload)
X, B = spec['X'], int(argv[1]) if len(argv) > 1 else 8192
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev) + 0.01)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
grad = None
if GRAD:
    by_id = {f['id']: f for f in spec['factors']}
    pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
    ukind = [2 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else (0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1) for j in topo.unary_factors]
    rs = np.random.RandomState(5)
    fb.set_features(rs.rand(X, X, 3), rs.rand(X, X, 3), rs.rand(X, spec['Vde'], 6), pair_phi, ukind)
    labels = np.tile(np.array([dict(zip(spec['var_ids'], spec['labels']))[v] for v in topo.var_ids]), (B, 1))
    fb.set_observations(labels, np.stack([rs.randint(0, spec['Vde'] if k == 2 else X, size=B) for k in ukind], axis=1))
    grad = (torch.empty(B, 3, dtype=torch.float64, device=dev), torch.empty(B, 6, dtype=torch.float64, device=dev))
_sweep = fb.sweep
fb.sweep = lambda r, **kw: _sweep(r, gradient=grad, keep_messages=not GRAD, **kw)
buf = torch.zeros((B // 64 + 1) * 12, dtype=torch.int64, device=dev)
masks = [(0, 'full kernel'), (1, '- main loop'), (2, '- final normalisation'), (4, '- message write-back'), (8, '- marginals'),
         (16, '- table loads'), (32, '- unary loads'), (1 | 2, '- loop, final norm'), (1 | 2 | 4 | 8, '- everything but the loads and the prologue'),
         (1 | 2 | 4 | 8 | 32, '- all but table loads + prologue'), (63, '- all of the above')]
times = {m: [] for m, _ in masks}
for _ in range(200):
    fb.sweep(roots, init=True, marginals=marg)
for rnd in range(8):
    for m, _ in masks:
        ffi.check(ffi.lib.mlbp_debug_lean_probe(m, None))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fb.sweep(roots, init=True, marginals=marg); e.record(); torch.cuda.synchronize()
        if rnd:
            times[m].append(s.elapsed_time(e))
base = sorted(times[0])[len(times[0]) // 2]
for m, name in masks:
    t = sorted(times[m])[len(times[m]) // 2]
    print('%-52s %.4f ms   delta %+.4f' % (name, t, t - base))
ffi.check(ffi.lib.mlbp_debug_lean_probe(0, buf.data_ptr()))
for _ in range(3):
    fb.sweep(roots, init=True, marginals=marg)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(-1, 12).astype(np.float64)
raw = raw[raw[:, 0] > 0]
names = ['issue loads (unary rows, tables, image), init LDS', 'unary normalisation + sync', 'constant products + sync', 'wait for tables',
         'main loop', 'read-out + final normalisation', 'sync + write-back issue', 'marginals issue', 'gradient epilogue + drain stores' if GRAD else 'drain stores']
d = np.diff(raw[:, :10], axis=1)
life = raw[:, 9] - raw[:, 0]
print('B = %d: wave-0 lifetime: mean %.0f cycles (min %.0f, max %.0f) over %d sampled workgroups' % (B, life.mean(), life.min(), life.max(), len(life)))
for i, n in enumerate(names):
    print('  %-50s %8.0f cycles  %5.1f %%' % (n, d[:, i].mean(), 100 * d[:, i].mean() / life.mean()))
