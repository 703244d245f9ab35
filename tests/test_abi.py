"""The C-ABI library loads without a GPU and exports every symbol include/mlbp.h declares; the
ctypes table in _ffi.py mirrors the header one to one.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from macaronicusermodeling_amd import _ffi

HEADER = os.path.join(ROOT, 'include', 'mlbp.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mlbp_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 30
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libmlbp.so does not export %s' % n
    assert sorted(_ffi.SIGNATURES) == names, (set(names) ^ set(_ffi.SIGNATURES))


def test_library_identity_and_host_side_errors():
    assert _ffi.lib.mlbp_arch() == b'gfx950'
    assert _ffi.lib.mlbp_version() >= 1
    assert _ffi.lib.mlbp_device_count() >= 0
    # host logic rejects bad input with a message, no GPU involved
    rc = _ffi.lib.mlbp_has_loops(None, 0)
    assert rc == _ffi.MLBP_EINVAL and 'NULL' in _ffi.last_error()
    with pytest.raises(_ffi.MlbpError):
        _ffi.check(rc)


def test_device_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    rc = _ffi.lib.mlbp_init_messages_f64(ctypes.c_void_p(8), 1, 4, None)
    assert rc == _ffi.MLBP_ENODEVICE and 'no CPU fallback' in _ffi.last_error()
    from macaronicusermodeling_amd.array_utils import c_array_utils as au
    import numpy as np
    with pytest.raises(_ffi.MlbpError):
        au.dense_dot(np.ones((2, 2)), np.ones((2, 2)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'macaronicusermodeling_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no oracle', ''), '%s mentions the oracle' % f
