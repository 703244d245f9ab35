"""Builds libmlbp.so (gfx950 only) in-tree with hipcc.  `python -m macaronicusermodeling_amd.build`.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the working-tree snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libmlbp.so')
SOURCES = ['mlbp_host.cpp', 'mlbp_sweep.hip', 'mlbp_lean.hip', 'mlbp_shared.hip', 'mlbp_gemm.hip', 'mlbp_prims.hip', 'mlbp_grad.hip']
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-Wall',
         '-Wno-unused-function']


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, 'mlbp_internal.h'), os.path.join(CSRC, 'mlbp_device.h'), os.path.join(PKG, '..', 'include', 'mlbp.h')]
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + FLAGS + ['-x', 'hip', '-c', path, '-o', obj]
            if verbose:
                cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        subprocess.check_call([hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
