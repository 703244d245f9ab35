"""GPU checks of the batch primitives around the sweep: the fixed-order column sums (batch_sgd_accumulate's reduction,
train_mp.py:405-424) and the potential construction (train_mp.py:220-255), against NumPy."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


@pytest.mark.parametrize('rows,cols', [(1, 1), (7, 3), (8192, 10), (8193, 9), (100003, 17), (300, 64)])
def test_sum_rows_any_shape(rows, cols):
    """mlbp_sum_rows_f64: column sums in a fixed order -- the same bits on every call -- equal to NumPy's to rounding; more than
    eight columns go through the register blocks, row counts that do not divide by the 64 slices leave ragged slices."""
    from macaronicusermodeling_amd import _ffi
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(rows + cols)
    a = rs.randn(rows, cols)
    t = torch.from_numpy(a).to(dev)
    out = torch.full((cols,), float('nan'), dtype=torch.float64, device=dev)
    out2 = torch.full((cols,), float('nan'), dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.mlbp_sum_rows_f64(t.data_ptr(), rows, cols, out.data_ptr(), _stream(dev)))
    _ffi.check(_ffi.lib.mlbp_sum_rows_f64(t.data_ptr(), rows, cols, out2.data_ptr(), _stream(dev)))
    assert torch.equal(out, out2)
    np.testing.assert_allclose(out.cpu().numpy(), a.sum(0), rtol=1e-11, atol=1e-9)


def test_sum_rows_of_three_arrays_with_the_count_appended():
    """mlbp_sum_rows_cat_f64: [sum g_ee | sum g_ed | sum log-posterior | count], the trainer's statistics vector."""
    from macaronicusermodeling_amd import _ffi
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(3)
    B = 5000
    a, b, c = rs.randn(B, 3), rs.randn(B, 6), rs.randn(B, 1)
    ta, tb, tc = (torch.from_numpy(x).to(dev) for x in (a, b, c))
    out = torch.empty(11, dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.mlbp_sum_rows_cat_f64(ta.data_ptr(), 3, tb.data_ptr(), 6, tc.data_ptr(), 1, B, 1, out.data_ptr(), _stream(dev)))
    want = np.concatenate([a.sum(0), b.sum(0), c.sum(0), [B]])
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-11, atol=1e-10)
    with pytest.raises(_ffi.MlbpError):
        _ffi.check(_ffi.lib.mlbp_sum_rows_cat_f64(ta.data_ptr(), 40, tb.data_ptr(), 30, None, 0, B, 0, out.data_ptr(), _stream(dev)))


def test_step_statistics_in_one_launch_equal_the_two_launch_form():
    """mlbp_step_statistics_f64 == mlbp_log_posterior_f64 then mlbp_sum_rows_cat_f64, bit for bit (LBP.py:247-259 with -inf ->
    -99.99, train_mp.py:405-424's sums); a label out of range is skipped and raises the status word."""
    from macaronicusermodeling_amd import _ffi
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(4)
    B, nv, X = 3001, 3, 64
    marg = rs.rand(B, nv, X); marg /= marg.sum(2, keepdims=True)
    labels = rs.randint(0, X, size=(B, nv)).astype(np.int32)
    marg[7, 1, labels[7, 1]] = 0.0                                # log 0 -> -99.99
    a, b = rs.randn(B, 3), rs.randn(B, 6)
    tm, tl, ta, tb = (torch.from_numpy(x).to(dev) for x in (marg, labels, a, b))
    lp1, lp2 = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev)
    out1, out2 = torch.empty(11, dtype=torch.float64, device=dev), torch.empty(11, dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.mlbp_log_posterior_f64(tm.data_ptr(), tl.data_ptr(), B, nv, X, lp1.data_ptr(), _stream(dev)))
    _ffi.check(_ffi.lib.mlbp_sum_rows_cat_f64(ta.data_ptr(), 3, tb.data_ptr(), 6, lp1.data_ptr(), 1, B, 1, out1.data_ptr(), _stream(dev)))
    _ffi.check(_ffi.lib.mlbp_step_statistics_f64(ta.data_ptr(), 3, tb.data_ptr(), 6, tm.data_ptr(), tl.data_ptr(), nv, X, B, lp2.data_ptr(),
                                                 out2.data_ptr(), _stream(dev)))
    assert torch.equal(lp1, lp2) and torch.equal(out1, out2)
    want_lp = np.log(np.take_along_axis(marg, labels[:, :, None].astype(np.int64), 2)[:, :, 0].clip(1e-300))
    want_lp[7, 1] = -99.99
    np.testing.assert_allclose(lp2.cpu().numpy(), want_lp.sum(1), rtol=1e-12)
    np.testing.assert_allclose(out2.cpu().numpy(), np.concatenate([a.sum(0), b.sum(0), [want_lp.sum()], [B]]), rtol=1e-11, atol=1e-9)
    _ffi.check(_ffi.lib.mlbp_step_statistics_f64(ta.data_ptr(), 3, tb.data_ptr(), 6, tm.data_ptr(), tl.data_ptr(), nv, X, B, None,
                                                 out1.data_ptr(), _stream(dev)))       # no per-graph output
    assert torch.equal(out1, out2) and _ffi.lib.mlbp_gradient_status() == 0
    tl[11, 0] = X
    _ffi.check(_ffi.lib.mlbp_step_statistics_f64(ta.data_ptr(), 3, tb.data_ptr(), 6, tm.data_ptr(), tl.data_ptr(), nv, X, B, lp2.data_ptr(),
                                                 out2.data_ptr(), _stream(dev)))
    assert _ffi.lib.mlbp_gradient_status() == 1


def test_potentials_of_several_feature_sets_and_parameter_vectors_in_one_launch():
    """mlbp_potentials_multi_f64 == exp(phi . theta) per job and repetition (train_mp.py:220-255), row-major and transposed
    outputs at their strides, identical to the one-job entry; bad job tables are refused."""
    from macaronicusermodeling_amd import _ffi
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(11)
    X, V, D = 16, 12, 5
    phi_a, phi_b = rs.randn(X, X, 3) * 0.3, rs.randn(X, V, 6) * 0.3
    th_a, th_b = rs.randn(D, 3), rs.randn(D, 6)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)          # noqa: E731
    pa, pb, ta, tb = d(phi_a), d(phi_b), d(th_a), d(th_b)
    pot_a = torch.zeros(D, X, X, dtype=torch.float64, device=dev)
    rows_per = X + V
    pot_t = torch.zeros(D * rows_per, X, dtype=torch.float64, device=dev)      # per repetition: [X rows of pot_a^T | V rows of pot_b^T]
    jobs = (_ffi.PotentialsJob * 2)()
    jobs[0].phi, jobs[0].theta, jobs[0].pot, jobs[0].pot_t = pa.data_ptr(), ta.data_ptr(), pot_a.data_ptr(), pot_t.data_ptr()
    jobs[0].theta_stride, jobs[0].pot_stride, jobs[0].pot_t_stride = 3, X * X, rows_per * X
    jobs[0].rows, jobs[0].cols, jobs[0].F = X, X, 3
    jobs[1].phi, jobs[1].theta, jobs[1].pot, jobs[1].pot_t = pb.data_ptr(), tb.data_ptr(), None, pot_t[X:].data_ptr()
    jobs[1].theta_stride, jobs[1].pot_stride, jobs[1].pot_t_stride = 6, 0, rows_per * X
    jobs[1].rows, jobs[1].cols, jobs[1].F = X, V, 6
    _ffi.check(_ffi.lib.mlbp_potentials_multi_f64(jobs, 2, D, _stream(dev)))
    got_a, got_t = pot_a.cpu().numpy(), pot_t.cpu().numpy().reshape(D, rows_per, X)
    for r in range(D):
        want_a, want_b = np.exp(phi_a.dot(th_a[r])), np.exp(phi_b.dot(th_b[r]))
        np.testing.assert_allclose(got_a[r], want_a, rtol=1e-13)
        np.testing.assert_allclose(got_t[r, :X], want_a.T, rtol=1e-13)
        np.testing.assert_allclose(got_t[r, X:], want_b.T, rtol=1e-13)
    one = torch.zeros(X, X, dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.mlbp_potentials_f64(pa.data_ptr(), ta[2].data_ptr(), X, X, 3, one.data_ptr(), None, _stream(dev)))
    assert torch.equal(one, pot_a[2])
    with pytest.raises(_ffi.MlbpError):
        _ffi.check(_ffi.lib.mlbp_potentials_multi_f64(jobs, 5, 1, _stream(dev)))
    jobs[1].pot_t = None
    with pytest.raises(_ffi.MlbpError):
        _ffi.check(_ffi.lib.mlbp_potentials_multi_f64(jobs, 2, 1, _stream(dev)))
