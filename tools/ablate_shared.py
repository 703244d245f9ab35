#!/usr/bin/env python3
"""Diagnostic (not product): runs the shared-table launch sequence of user_k3 in the trainer's layout N times against a -DMLBP_STAMPS
-DMLBP_STAMPS_LIGHT build (tools/bin/libmlbp_ablate.so, built in the container: `python tools/ablate_shared.py --build`) with one
ablation mask (argv[1]; g_sh_ablate in mlbp_shared.hip); meant to run under `rocprofv3 --kernel-trace --stats`, whose per-kernel
averages then say what the removed part costs."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
lib = os.path.join(ROOT, 'tools', 'bin', 'libmlbp_ablate.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
if '--build' in sys.argv:
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-DMLBP_STAMPS', '-DMLBP_STAMPS_LIGHT',
                           '-shared', '-x', 'hip'] + [os.path.join(csrc, f) for f in ('mlbp_host.cpp', 'mlbp_sweep.hip', 'mlbp_lean.hip', 'mlbp_shared.hip', 'mlbp_gemm.hip',
                                                                                     'mlbp_prims.hip', 'mlbp_grad.hip')] + ['-o', lib])
    raise SystemExit(0)
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_set_shared_stamp_buffer.argtypes = [C.c_void_p, C.c_int]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

mask = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8192
spec, roots, sweeps, seed = bench.workload_spec('user_k3_shared')
X = spec['X']
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
by_id = {f['id']: f for f in spec['factors']}
which = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
fb.set_pair_tables(torch.rand(2, X, X, dtype=torch.float64, device=dev) + 0.01, np.tile(np.array(which), (B, 1)))
fb.set_unary_tables(torch.rand(192, X, dtype=torch.float64, device=dev) + 0.01, np.random.RandomState(0).randint(0, 192, size=(B, topo.U)))
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
assert ffi.lib.mlbp_debug_set_shared_stamp_buffer(None, mask) == 0
if '--gradient' in sys.argv:            # the trainer's step: gradient as the sweep kernel's epilogue
    import cases as CS
    from macaronicusermodeling_amd.train import UserGraphTrainer
    spec = CS.user_spec(10, [1, 4, 7], X, 64, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = CS.reference_planes(CS.make_inputs(spec, 5))
    rs = np.random.RandomState(0)
    tr = UserGraphTrainer(spec, rs.randint(0, X, size=(B, topo.n_vars)), rs.randint(0, 64, size=(B, topo.U)), inputs['phi_en_en'],
                          inputs['phi_en_en_w1'], inputs['phi_en_de'], inputs['theta_en_en'], inputs['theta_en_de'])
    for _ in range(300):
        tr.local_statistics()
else:
    for _ in range(300):
        fb.sweep(roots, init=True, marginals=marg, keep_messages=False)
torch.cuda.synchronize()
assert ffi.lib.mlbp_last_sweep_kernel() == 3
