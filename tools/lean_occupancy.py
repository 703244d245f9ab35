#!/usr/bin/env python3
"""Diagnostic (not product): the lean X = 64 kernel at fewer workgroups (= graphs in flight) per CU.  Builds copies of
libmlbp.so with -DMLBP_LEAN_EXTRA_LDS=<bytes> (the extra LDS is never touched; it only lowers the occupancy the hardware
grants) and times bench.py's default sweep call: if the launch time follows the number of graphs in flight, a workgroup
that interleaves two graphs in its waves at HALF the workgroups per CU has nothing to gain."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'lean_occupancy')
os.makedirs(OUT, exist_ok=True)
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
from macaronicusermodeling_amd import build as B_  # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
workload = sys.argv[2] if len(sys.argv) > 2 else 'user_k3'
lib = os.path.join(OUT, 'libmlbp_extra%d.so' % extra)
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math',
                       '-DMLBP_LEAN_EXTRA_LDS=%d' % extra, '-shared', '-x', 'hip'] + [os.path.join(csrc, f) for f in B_.SOURCES] + ['-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

spec, roots, sweeps, seed = bench.workload_spec(workload)
X, B = spec['X'], 8192 if workload.startswith('user') else 1024
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev) + 0.01)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
for _ in range(300):
    fb.sweep(roots, init=True, marginals=marg)
torch.cuda.synchronize()
ts = []
for rnd in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fb.sweep(roots, init=True, marginals=marg)
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 20)
assert ffi.lib.mlbp_last_sweep_kernel() == 7
print('%s B=%d extra LDS %6d bytes: %.4f ms per sweep call (median of 5 x 20) | %s' % (workload, B, extra, sorted(ts)[2], ffi.lib.mlbp_last_error().decode()[:120]))
