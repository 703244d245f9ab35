#!/usr/bin/env python3
"""Diagnostic (not product): builds a -DMLBP_STAMPS copy of libmlbp.so under gpurun_out/ and prints
the share of shader-clock cycles each phase of the EXACT X = 64 kernel (sweep_x64_fused_kernel, variant 3) takes (wave 0 of
the first 64 workgroups; the lean kernel has tools/lean_probe.py, the shared-table kernel tools/stamp_shared.py).  Shares only -- the stamped build is slower than the shipped one."""
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
                       '-DMLBP_STAMPS', '-shared', '-x', 'hip', os.path.join(csrc, 'mlbp_host.cpp'),
                       os.path.join(csrc, 'mlbp_sweep.hip'), os.path.join(csrc, 'mlbp_lean.hip'), os.path.join(csrc, 'mlbp_shared.hip'), os.path.join(csrc, 'mlbp_gemm.hip'), os.path.join(csrc, 'mlbp_prims.hip'),
                       os.path.join(csrc, 'mlbp_grad.hip'), '-o', lib])
import macaronicusermodeling_amd._ffi as ffi  # noqa: E402
ffi.LIB_PATH = lib
ffi.lib = ffi._load()
ffi.lib.mlbp_debug_set_stamp_buffer.argtypes = [C.c_void_p]
import torch  # noqa: E402
import bench  # noqa: E402
from macaronicusermodeling_amd.batch import FactorGraphBatch  # noqa: E402
from macaronicusermodeling_amd.topology import GraphTopology  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'user_k3'
variant = 3          # the exact kernel on every graph (variants 0 and 2, the kernels this tool once stamped, were retired in round 3)
spec, roots, sweeps, seed = bench.workload_spec(workload)
X, B = spec['X'], 8192
topo = GraphTopology.from_spec(spec)
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev) + 0.01)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
buf = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
ffi.check(ffi.lib.mlbp_debug_set_stamp_buffer(buf.data_ptr()))
ffi.check(ffi.lib.mlbp_set_sweep_variant(variant))
for _ in range(3):
    fb.sweep(roots, init=True)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(64, 8)
print('debug bits of phase 2 (wg0): slow=%d neg=%d inf=%d allzero=%d' % ((raw[0,2]>>40)&15, (raw[0,2]>>44)&15, (raw[0,2]>>48)&15, (raw[0,2]>>52)&15))
t = (raw & ((1<<40)-1)).astype(float)
names = ['prologue', 'op header', 'var product', 'var normalise', 'pair partials', 'barrier', 'gather partials', 'pair normalise']
tot = t.sum(1).mean()
print('workload %s variant %d: mean cycles per workgroup %.0f' % (workload, variant, tot))
for i, n in enumerate(names):
    print('  %-16s %9.0f cycles  %5.1f %%' % (n, t[:, i].mean(), 100 * t[:, i].mean() / tot))
