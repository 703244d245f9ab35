#!/usr/bin/env python3
"""Diagnostic (not product): shader-clock stamps of shared_prepare_kernel (every load waited for at each stamp, so the phases are
the dependent rounds of memory latency a wave goes through) on the trainer's step -- a -DMLBP_STAMPS build under gpurun_out/."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
OUT = os.path.join(ROOT, 'gpurun_out', 'stamps')
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, 'libmlbp_stamps_prepare.so')
csrc = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
from macaronicusermodeling_amd import build as B_  # noqa: E402
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-DMLBP_STAMPS', '-shared', '-x', 'hip'] +
                      [os.path.join(csrc, f) for f in B_.SOURCES] + ['-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_set_prepare_stamp_buffer.argtypes = [C.c_void_p]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cases as CS  # noqa: E402
from macaronicusermodeling_amd.train import UserGraphTrainer  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

B, X = 8192, 64
spec = CS.user_spec(10, [1, 4, 7], X, 64, seed=1)
topo = GraphTopology.from_spec(spec)
inputs = CS.reference_planes(CS.make_inputs(spec, 5))
rs = np.random.RandomState(0)
tr = UserGraphTrainer(spec, rs.randint(0, X, size=(B, topo.n_vars)), rs.randint(0, 64, size=(B, topo.U)), inputs['phi_en_en'],
                      inputs['phi_en_en_w1'], inputs['phi_en_de'], inputs['theta_en_en'], inputs['theta_en_de'])
buf = torch.zeros(64 * 4 * 8, dtype=torch.int64, device='cuda:0')
assert ffi.lib.mlbp_debug_set_prepare_stamp_buffer(buf.data_ptr()) == 0
for _ in range(5):
    tr.local_statistics()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(64 * 4, 8).astype(float)
staged = t[:, 6].mean() > 0
d = np.diff(t[:, :7 if staged else 5], axis=1)
names = ['row indices arrive', 'table copied into LDS', 'table jobs + barrier', 'unary gradient gathers', 'entries loop (rows from LDS), sums, tile stores', 'verdict, gradient sums, stores drained'] if staged else ['row indices arrive', "unary gradient's gathers arrive", 'rows, products, sums, tile stores', 'verdict, gradient sums, stores drained']
print('shared_prepare_kernel, waves of the first 64 blocks, ticks (about ns) between stamps; every load is waited for at a stamp:')
for i, n in enumerate(names):
    print('  %-42s mean %7.0f  min %7.0f  max %7.0f' % (n, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
print('  total %.0f' % d.sum(1).mean())
