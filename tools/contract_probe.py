#!/usr/bin/env python3
"""Diagnostic (not product): times the shared-table contraction launch of bench.py's ring8_x512_shared workload in the
shipped build and in a -DMLBP_CONTRACT_NOLOOP build (main loop cut to 4 of its 64 steps: prologue + epilogue only; wrong
results, only the time is read)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'contract_probe')
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, 'libmlbp_noloop.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
from macaronicusermodeling_amd import build as B_  # noqa: E402
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
if '--noloop' in sys.argv or extra:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math',
                           *((['-DMLBP_CONTRACT_NOLOOP'] if '--noloop' in sys.argv else []) + extra), '-shared', '-x', 'hip'] + [os.path.join(csrc, f) for f in B_.SOURCES] + ['-o', lib])
    import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
    ffi.LIB_PATH = lib
    ffi.lib = ffi._load()
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

f32 = '--f32' in sys.argv
spec, roots, sweeps, seed = bench.workload_spec('ring8_x512_shared')
X, B = spec['X'], 8192
for a_ in sys.argv[1:]:
    if a_.startswith('--X='):
        X = int(a_[4:])          # other table sizes of the same ring (128, 256, 384, 512)
    if a_.startswith('--B='):
        B = int(a_[4:])
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(topo.P, X, X, dtype=torch.float64, device=dev) + 0.01, np.tile(np.arange(topo.P), (B, 1)),
                   dtype=torch.float32 if f32 else torch.float64)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
ts = []
for rnd in range(4):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fb.sweep(roots, init=True); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
n_upd = 16 * len(roots)
print('X=%d B=%d %s%s: %.3f ms per %d-sweep call = %.1f us per update = %.1f TFLOP/s' % (X, B, 'noloop ' if '--noloop' in sys.argv else 'full ', 'f32' if f32 else 'f64', min(ts), len(roots), 1e3 * min(ts) / n_upd, 2.0 * X * X * B * n_upd / (min(ts) * 1e-3) / 1e12))
