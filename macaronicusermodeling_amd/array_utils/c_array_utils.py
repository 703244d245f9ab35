"""MI355X-backed drop-in for the reference's Cython module `array_utils.c_array_utils`.

Same function names, argument meaning, return shapes, in-place behaviour and error types as the
reference (file:line cited per function; errors recorded from the real module are in
tests/golden/MANIFEST.json).  NumPy arrays in, NumPy arrays out -- but every floating-point
operation runs in a HIP kernel behind libmlbp.so (include/mlbp.h): arrays are staged to the GPU
through torch tensors, the C ABI is called, results are copied back.  There is NO CPU fallback:
without an MI355X these functions raise `MlbpError(MLBP_ENODEVICE)`.

One call = a few launches and two PCIe copies, so this surface is for drop-in use and parity
checking; the performance path is the fused, batched sweep (macaronicusermodeling_amd.batch).
"""
import ctypes as C
import itertools  # noqa: F401  (names the reference module exposes, LBP users may poke at them)
import time  # noqa: F401
import warnings  # noqa: F401

import numpy as np
import torch
from scipy import sparse  # noqa: F401  (imported, unused, in the reference too: pyx:7)

from .. import _ffi

K_SPARSE = 100   # `cdef int K = 100`, c_array_utils.pyx:118,194


def _dev():
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _up(a):
    """NumPy array -> device tensor (float64), keeping strides of transposed views."""
    dev = _dev()
    if dev is None:
        raise _ffi.MlbpError(_ffi.MLBP_ENODEVICE, 'no MI355X visible: c_array_utils has no CPU fallback')
    a = np.asarray(a)
    if a.dtype != np.float64:
        a = a.astype(np.float64)
    if any(s < 0 for s in a.strides):
        a = np.ascontiguousarray(a)
    return torch.from_numpy(a).to(dev)


def _idx_up(idx, limit):
    idx = np.asarray(idx)
    if idx.ndim != 1:
        raise ValueError('Buffer has wrong number of dimensions (expected 1, got %d)' % idx.ndim)
    if idx.dtype.kind not in 'iu':
        raise ValueError("Buffer dtype mismatch, expected 'int64_t' but got '%s'" % idx.dtype.name)
    if idx.size and (idx.min() < -limit or idx.max() >= limit):
        raise IndexError('index out of bounds for axis with size %d' % limit)
    idx = np.where(idx < 0, idx + limit, idx).astype(np.int32)
    return torch.from_numpy(np.ascontiguousarray(idx)).to(_dev())


def _typed(a, name='buffer'):
    """The reference's typed `np.ndarray[np.float64_t, ndim=2]` arguments reject everything else
    with ValueError (c_array_utils.pyx:90,93,108-111,117,193)."""
    if not isinstance(a, np.ndarray):
        raise TypeError('Argument has incorrect type (expected numpy.ndarray, got %s)' % type(a).__name__)
    if a.dtype != np.float64:
        got = {'float32': 'float', 'int64': 'long', 'int32': 'int'}.get(a.dtype.name, a.dtype.name)
        raise ValueError("Buffer dtype mismatch, expected 'float64_t' but got '%s'" % got)
    if a.ndim != 2:
        raise ValueError('Buffer has wrong number of dimensions (expected 2, got %d)' % a.ndim)
    return a


def _multiply(m1, m2):
    a, b = np.broadcast_arrays(np.asarray(m1), np.asarray(m2))
    shape = a.shape
    if a.size == 0:
        return np.zeros(shape, dtype=np.float64)
    ta, tb = _up(np.ascontiguousarray(a)), _up(np.ascontiguousarray(b))
    out = torch.empty_like(ta)
    _ffi.check(_ffi.lib.mlbp_pointwise_multiply_f64(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), ta.numel(), 0,
                                                    _stream(ta.device)))
    return out.cpu().numpy().reshape(shape)


def pointwise_multiply(m1, m2):
    """c_array_utils.pyx:12-16 (`np.multiply`, untyped)."""
    return _multiply(m1, m2)


def dense_pointwise_multiply(m1, m2):
    """c_array_utils.pyx:93-94 (typed 2-D float64)."""
    return _multiply(_typed(m1), _typed(m2))


def normalize(m1):
    """c_array_utils.pyx:29-40: total > 0 -> a NEW array m1 / total; otherwise m1 is zero-filled IN
    PLACE and returned (same object)."""
    t = _up(np.ascontiguousarray(m1))
    out = torch.empty_like(t)
    pos = torch.zeros(1, dtype=torch.int32, device=t.device)
    _ffi.check(_ffi.lib.mlbp_normalize_f64(t.data_ptr(), out.data_ptr(), 1, t.numel(), _ffi.NORM_ZERO,
                                           pos.data_ptr(), _stream(t.device)))
    if int(pos.item()):
        return out.cpu().numpy().reshape(np.shape(m1))
    m1.fill(0)
    return m1


def dense_dot(m1, m2):
    """c_array_utils.pyx:90-91: `m1.dot(m2)` for 2-D float64 arrays; strided views such as `msg.m.T`
    (LBP.py:518) are accepted and multiplied in place on the device without a transposing copy."""
    m1, m2 = _typed(m1), _typed(m2)
    if m1.shape[1] != m2.shape[0]:
        raise ValueError('shapes %s and %s not aligned: %d (dim 1) != %d (dim 0)'
                         % (m1.shape, m2.shape, m1.shape[1], m2.shape[0]))
    M, K = m1.shape
    N = m2.shape[1]
    a, b = _up(m1), _up(m2)
    c = torch.empty(M, N, dtype=torch.float64, device=a.device)
    _ffi.check(_ffi.lib.mlbp_dense_dot_f64(1, M, K, N, a.data_ptr(), 0, a.stride(0), a.stride(1),
                                           b.data_ptr(), 0, b.stride(0), b.stride(1),
                                           c.data_ptr(), 0, N, _stream(a.device)))
    return c.cpu().numpy()


def _topk(t_vec, stride, n, K=K_SPARSE):
    """Device top-K -> device int32 [K] (descending value order).  K > n raises the ValueError NumPy's
    argpartition raises in the reference."""
    if K > n:
        raise ValueError('kth(=%d) out of bounds (%d)' % (K - 1, n))
    idx = torch.empty(K, dtype=torch.int32, device=t_vec.device)
    _ffi.check(_ffi.lib.mlbp_topk_f64(t_vec.data_ptr(), stride, n, K, idx.data_ptr(), _stream(t_vec.device)))
    return idx


def sparse_vec_mat_dot(vec, mat):
    """c_array_utils.pyx:193-205.  Row vector (1,X) -> 1-D (X,) result; column vector (X,1) ->
    (X,1) result; only the K=100 largest entries of `vec` contribute."""
    vec, mat = _typed(vec), _typed(mat)
    v, m = _up(vec), _up(mat)
    if vec.shape[0] == 1:
        n = vec.shape[1]
        idx = _topk(v, v.stride(1), n)
        n_out = mat.shape[1]
        out = torch.empty(n_out, dtype=torch.float64, device=v.device)
        _ffi.check(_ffi.lib.mlbp_sparse_vec_mat_dot_f64(v.data_ptr(), v.stride(1), m.data_ptr(), m.stride(0),
                                                        m.stride(1), n_out, idx.data_ptr(), K_SPARSE, 1,
                                                        out.data_ptr(), _stream(v.device)))
        return out.cpu().numpy()
    n = vec.shape[0]
    idx = _topk(v, v.stride(0), n)
    n_out = mat.shape[0]
    out = torch.empty(n_out, 1, dtype=torch.float64, device=v.device)
    _ffi.check(_ffi.lib.mlbp_sparse_vec_mat_dot_f64(v.data_ptr(), v.stride(0), m.data_ptr(), m.stride(0), m.stride(1),
                                                    n_out, idx.data_ptr(), K_SPARSE, 0, out.data_ptr(),
                                                    _stream(v.device)))
    return out.cpu().numpy()


def sparse_dot(m1, m2):
    """c_array_utils.pyx:117-129 -> (out (n,n), m1_idx (K,), m2_idx (K,)); index arrays are platform
    ints (int64) like the reference; their order within the set is unspecified there, descending
    value here."""
    m1, m2 = _typed(m1), _typed(m2)
    assert m1.shape[0] == m2.shape[1]
    assert m1.shape[1] == m2.shape[0] == 1
    n = m1.shape[0]
    c, r = _up(np.ascontiguousarray(m1)), _up(np.ascontiguousarray(m2))
    ci, ri = _topk(c, 1, n), _topk(r, 1, n)
    out = torch.empty(n, n, dtype=torch.float64, device=c.device)
    _ffi.check(_ffi.lib.mlbp_sparse_dot_f64(c.data_ptr(), r.data_ptr(), n, ci.data_ptr(), ri.data_ptr(), K_SPARSE,
                                            out.data_ptr(), _stream(c.device)))
    return out.cpu().numpy(), ci.cpu().numpy().astype(np.int64), ri.cpu().numpy().astype(np.int64)


def sparse_pointwise_multiply(sparse_m, c_idx, r_idx, dense_m):
    """c_array_utils.pyx:108-114."""
    sparse_m, dense_m = _typed(sparse_m), _typed(dense_m)
    if sparse_m.shape != dense_m.shape:
        raise ValueError('operands could not be broadcast together with shapes %s %s' % (sparse_m.shape, dense_m.shape))
    rows, cols = dense_m.shape
    s, d = _up(np.ascontiguousarray(sparse_m)), _up(np.ascontiguousarray(dense_m))
    ci, ri = _idx_up(c_idx, rows), _idx_up(r_idx, cols)
    out = torch.empty(rows, cols, dtype=torch.float64, device=s.device)
    _ffi.check(_ffi.lib.mlbp_sparse_pointwise_multiply_f64(s.data_ptr(), d.data_ptr(), rows, cols, ci.data_ptr(),
                                                           len(ci), ri.data_ptr(), len(ri), out.data_ptr(),
                                                           _stream(s.device)))
    return out.cpu().numpy()


def sparse_normalize(m1, c_idx, r_idx):
    """c_array_utils.pyx:23-26: the (c_idx x r_idx) block is divided by its own sum IN PLACE and m1
    itself is returned."""
    if not isinstance(m1, np.ndarray) or m1.ndim != 2:
        raise ValueError('sparse_normalize needs a 2-D array')
    rows, cols = m1.shape
    t = _up(np.ascontiguousarray(m1))
    ci, ri = _idx_up(c_idx, rows), _idx_up(r_idx, cols)
    scratch = torch.empty(1, dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib.mlbp_sparse_normalize_f64(t.data_ptr(), cols, ci.data_ptr(), len(ci), ri.data_ptr(), len(ri),
                                                  scratch.data_ptr(), _stream(t.device)))
    m1[...] = t.cpu().numpy()
    return m1


# ---- the functions of the reference module that NOTHING in the reference calls -----------------------------------------
# (c_array_utils.pyx:18-20, 43-75, 96-105, 132-190.)  Not on the hot path and not accelerated: small
# host-side NumPy bodies with the reference's argument meaning, in-place behaviour, return types and errors, so that code
# which imports the module finds every name doing what it did.  Pinned on outputs of the reference's own module
# (tests/golden/au_dormant_functions.npz, generated by make_golden.py; tests/test_au_dormant.py).  Top-K selections: the K
# largest entries, ties by lower index (the rule of mlbp_topk_f64; np.argpartition leaves the order open).
def _top_k(flat, k):
    """Indices of the k largest entries of a 1-D array (ties: lower index first), ascending by index."""
    order = np.lexsort((np.arange(flat.size), -flat))
    return np.sort(order[:k])


def clip(m1):
    """c_array_utils.pyx:18-20: entries below 1e-100 become 0, IN PLACE; returns its argument."""
    m1[m1 < 1.0e-100] = 0.0
    return m1


def induce_s_pointwise_multiply_clip(d1, d2):
    """c_array_utils.pyx:43-50: zeros like d2 except at the K = 100 largest cells of d1, where d1 * d2."""
    d1, d2 = np.asarray(d1), np.asarray(d2)
    if __debug__:
        assert np.shape(d1) == np.shape(d2)
    if K_SPARSE >= d1.size:                      # `(-d1).argpartition(K, axis=None)`
        raise ValueError('kth(=%d) out of bounds (%d)' % (K_SPARSE, d1.size))
    x, y = np.unravel_index(_top_k(d1.reshape(-1), K_SPARSE), d1.shape)
    result = np.zeros_like(d2)
    result[x, y] = d1[x, y] * d2[x, y]
    return result


def induce_s(m1):
    """c_array_utils.pyx:53-63: a column vector with everything but its K = 100 largest entries zeroed (a NEW array); a
    vector of fewer than K entries comes back as it is (the same object)."""
    if __debug__:
        assert np.shape(m1)[1] == 1
    if K_SPARSE > np.size(m1):
        return m1
    if K_SPARSE >= np.size(m1):
        raise ValueError('kth(=%d) out of bounds (%d)' % (K_SPARSE, np.size(m1)))
    x, y = np.unravel_index(_top_k(np.asarray(m1).reshape(-1), K_SPARSE), np.shape(m1))
    new_m1 = np.zeros_like(m1)
    new_m1[x, y] = m1[x, y]
    return new_m1


def induce_s_mutliply_clip(s1, d2):
    """c_array_utils.pyx:66-75 (the reference's spelling): d2[:, idx] . s1[idx] over the K = 100 entries of s1 largest in
    magnitude."""
    if __debug__:
        assert np.shape(d2)[0] < np.shape(d2)[1]
        assert np.shape(s1)[0] == np.shape(d2)[1] and np.shape(s1)[1] == 1
    n = np.size(s1)
    if K_SPARSE > n:                             # `np.argpartition(s1_abs, -K)`
        raise ValueError('kth(=%d) out of bounds (%d)' % (n - K_SPARSE, n))
    idx = _top_k(np.abs(np.asarray(s1)).reshape(n), K_SPARSE)
    return np.asarray(d2)[:, idx].dot(np.asarray(s1)[idx, :])


def make_sparse_and_dot(m1, m2):
    """c_array_utils.pyx:96-105: {(x, y): m1[x, 0] * m2[0, y]} over the K = 100 largest entries of each."""
    a, b = np.reshape(m1, np.size(m1)), np.reshape(m2, np.size(m2))
    for v in (a, b):
        if K_SPARSE > v.size:
            raise ValueError('kth(=%d) out of bounds (%d)' % (v.size - K_SPARSE, v.size))
    xs, ys = _top_k(a, K_SPARSE), _top_k(b, K_SPARSE)
    return {(x, y): m1[x, 0] * m2[0, y] for x, y in itertools.product(xs, ys)}


def sparse_multiply_and_normalize(s_m1, m2):
    """c_array_utils.pyx:132-142: s_m1 = {(x, y): v}; returns (zeros like m2 with m2[x, y] * v at those cells, normalised
    over them; the same values as a dict).  The total is accumulated in the dict's iteration order, like the reference."""
    m2_z = np.zeros_like(m2)
    m2_d = {}
    n = 0.0
    for (x, y), v in s_m1.items():
        m2_z[x, y] = m2[x, y] * v
        n += m2_z[x, y]
    for x, y in s_m1:
        m2_z[x, y] = m2_z[x, y] / n
        m2_d[x, y] = m2_z[x, y]
    return m2_z, m2_d


def sd_matrix_multiply(s1, d2):
    """c_array_utils.pyx:145-146: `s1.dot(d2)` (a scipy sparse matrix times a dense one)."""
    return s1.dot(d2)


def ss_matix_multiply(s1, s2):
    """c_array_utils.pyx:153-154 (the reference's spelling): `s1.dot(s2)`, sparse times sparse."""
    return s1.dot(s2)


def make_adapt_phi(phi, num_adaptations):
    """c_array_utils.pyx:157-161: [phi | 0 | ... | 0], num_adaptations zero copies of phi's width behind it."""
    adapt_phi = np.zeros((np.shape(phi)[0], np.shape(phi)[1] * (num_adaptations + 1)))
    adapt_phi[:, :np.shape(phi)[1]] = phi
    return adapt_phi


def set_adaptation(f_size, adapt_phi, active_adaptations):
    """c_array_utils.pyx:164-173: block i (columns i f .. i f + f) := block 0 for every active i, IN PLACE."""
    f = f_size
    for i in active_adaptations:
        adapt_phi[:, i * f:i * f + f] = adapt_phi[:, :f]
    return adapt_phi


def set_adaptation_off(f_size, adapt_phi, active_adaptations):
    """c_array_utils.pyx:176-184: block i := 0 for every listed i, IN PLACE."""
    f = f_size
    for i in active_adaptations:
        adapt_phi[:, i * f:i * f + f] = 0
    return adapt_phi


def set_original(phi, adapt_phi):
    """c_array_utils.pyx:187-190: block 0 := phi, IN PLACE."""
    adapt_phi[:, :np.shape(phi)[1]] = phi
    return adapt_phi


def induce_s_multiply_threshold(s1, d2):
    """c_array_utils.pyx:78-79: raises in the reference too."""
    raise NotImplementedError('do not use it seems very slow..')


def sd_pointwise_multiply(s1, d2):
    """c_array_utils.pyx:149-150: raises in the reference too."""
    raise NotImplementedError('not implemented pointwise multiply for sparse-dense matrix')
