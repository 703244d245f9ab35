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
GRAD = '--gradient' in sys.argv        # the trainer's step: gradient as the sweep kernel's epilogue
if GRAD:
    import cases as CS
    from macaronicusermodeling_amd.train import UserGraphTrainer
    spec = CS.user_spec(10, [1, 4, 7], X, 64, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = CS.make_inputs(spec, 5)
    if '--random-planes' not in sys.argv:        # the reference's tensors: [pmi, 0, 1] and [pmi, pmi_w1, 1] (train_mp.py:600-606)
        inputs = CS.reference_planes(inputs)
    rs = np.random.RandomState(0)
    tr = UserGraphTrainer(spec, rs.randint(0, X, size=(B, topo.n_vars)), rs.randint(0, 64, size=(B, topo.U)), inputs['phi_en_en'],
                          inputs['phi_en_en_w1'], inputs['phi_en_de'], inputs['theta_en_en'], inputs['theta_en_de'])
names = ['A: tables, bundles, product tiles', 'C: partition choice, fragment loads, barriers', 'loop: quarters: tile reads + products',
         'loop: totals of the product, result store, column sum', 'loop: waiting at the barrier (incl. bundles this wave sits out)', 'epilogue',
         'loop: bundle decode + source totals -> scale', 'loop: quarters: MFMA issue (4 dependent per quarter)',
         'gradient: barrier + messages back into tiles', 'gradient: fragment fetches + MFMAs + partial sums', 'gradient: per-graph pair terms (label gathers)', 'gradient: unary gathers + stores']
KEEP = '--no-writeback' not in sys.argv
EXTRA = ((32, 'gradient WITHOUT the message rows'), (64, 'gradient WITHOUT the items'), (128, 'gradient WITHOUT the unary gathers'), (256, 'gradient WITHOUT the fragment fetches'), (480, 'gradient WITHOUT all four')) if GRAD else ()
for mask, what in ((0, 'nothing removed'),) + EXTRA + ( (1, 'WITHOUT the MFMAs'), (2, 'WITHOUT the tile reads'), (3, 'WITHOUT MFMAs and tile reads'), (4, 'WITHOUT the result stores'), (8, 'prepare kernel WITHOUT its row loads'), (16, 'prepare kernel WITHOUT its copy-out'), (24, 'prepare kernel WITHOUT both')):
    if GRAD and mask and mask < 32:
        continue
    if mask and '--ablate' not in sys.argv:
        break
    buf = torch.zeros(64 * 8 * 12, dtype=torch.int64, device=dev)
    assert ffi.lib.mlbp_debug_set_shared_stamp_buffer(buf.data_ptr(), mask) == 0
    run = (lambda: tr.local_statistics()) if GRAD else (lambda: fb.sweep(roots, init=True, marginals=marg, keep_messages=KEEP))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        run()
    e.record(); torch.cuda.synchronize()
    assert ffi.lib.mlbp_last_sweep_kernel() == 3
    if mask == 0:
        print(ffi.lib.mlbp_last_error().decode())
    t = buf.cpu().numpy().reshape(64, 8, 12).astype(float)          # [workgroup][wave][phase]
    us = s.elapsed_time(e) / 5 * 1e3
    tot = t.sum(2)
    print('B=%d %s: %.1f us per launch sequence (stamped build); ticks per wave lifetime: mean %.0f, min %.0f, max %.0f = %.2f ticks per ns of the sequence'
          % (B, what, us, tot.mean(), tot.min(), tot.max(), tot.mean() / (us * 1e3)))
    for w in range(8 if '--light' not in sys.argv else 0):
        print('  wave %d (half %d, rows %2d..): ' % (w, w >> 2, 16 * (w & 3)) + '  '.join('%.0f' % t[:, w, i].mean() for i in range(12)))
print('  phases: ' + ' | '.join(names))
