#!/usr/bin/env python3
"""Diagnostic (not product): builds a -DMLBP_STAMPS copy of libmlbp.so under gpurun_out/ and prints the
shader-clock cycles each phase of the shared-table (MFMA) sweep kernel takes (wave 0 of the first 64
workgroups)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'stamps')
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, 'libmlbp_stamps.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math',
                       '-DMLBP_STAMPS', *(['-DMLBP_STAMPS_LIGHT'] if '--light' in sys.argv else []), '-shared', '-x', 'hip'] +
                      [os.path.join(csrc, f) for f in ('mlbp_host.cpp', 'mlbp_sweep.hip', 'mlbp_lean.hip', 'mlbp_shared.hip', 'mlbp_gemm.hip', 'mlbp_prims.hip',
                                                       'mlbp_grad.hip')] + ['-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_set_shared_stamp_buffer.argtypes = [C.c_void_p, C.c_int]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

spec, roots, sweeps, seed = bench.workload_spec('user_k3_shared')
X, B = spec['X'], int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
by_id = {f['id']: f for f in spec['factors']}
which = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
fb.set_pair_tables(torch.rand(2, X, X, dtype=torch.float64, device=dev) + 0.01, np.tile(np.array(which), (B, 1)))
if '--shared-unary' in sys.argv:      # the trainer's layout: every unary table is one of 192 rows of the transposed pots (L2-resident)
    fb.set_unary_tables(torch.rand(192, X, dtype=torch.float64, device=dev) + 0.01, np.random.RandomState(0).randint(0, 192, size=(B, topo.U)))
else:
    fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
names = ['A: indices/image/init', 'B: table fragment loads issued', 'C: unary -> products + barrier', 'loop: tile reads + products',
         'loop: mfma + store', 'loop: barrier', 'epilogue', 'loop: column sum + v->f stores']
ABL = ['unary write-back stores', 'transposed tile writes', 'wave_sum', 'unary loads', 'v->f stores', 'column_sum', 'mfma',
       'second source tiles']
KEEP = '--no-writeback' not in sys.argv
masks = [0] + [1 << i for i in range(8)] if '--ablate' in sys.argv else [0]
for mask in masks:
    buf = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
    assert ffi.lib.mlbp_debug_set_shared_stamp_buffer(buf.data_ptr(), mask) == 0
    for _ in range(3):
        fb.sweep(roots, init=True, marginals=marg, keep_messages=KEEP)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fb.sweep(roots, init=True, marginals=marg, keep_messages=KEEP)
    e.record(); torch.cuda.synchronize()
    assert ffi.lib.mlbp_last_sweep_kernel() == 3
    if mask == 0:
        print(ffi.lib.mlbp_last_error().decode())
    t = buf.cpu().numpy().reshape(64, 8).astype(float)
    tot = t.sum(1).mean()
    what = 'nothing removed' if mask == 0 else 'WITHOUT ' + ABL[mask.bit_length() - 1]
    us = s.elapsed_time(e) / 5 * 1e3
    print('B=%d %s: %.1f us per launch (stamped build); ticks per workgroup %.0f (min %.0f, max %.0f) = %.2f ticks per ns of the launch' % (B, what, us, tot, t.sum(1).min(), t.sum(1).max(), tot / (us * 1e3)))
    print('   ' + '  '.join('%s %.0f' % (n.split(':')[0] if i < 3 else n[6:], t[:, i].mean()) for i, n in enumerate(names)))
