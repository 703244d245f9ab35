"""Pins the CPU oracle (oracle/) against fixtures produced by RUNNING the reference
(tests/golden/make_golden.py).  Integer work bit-exact; float work to 1e-12 relative (same NumPy
calls in the same order; BLAS summation order inside `dot` is the only freedom)."""
import json
import os

import numpy as np
import pytest

import cases as C
from conftest import GOLDEN, load_golden
from oracle import array_oracle as au
from oracle import lbp_oracle as O

RTOL = 1e-12


def _sched_array(pairs):
    return np.array([[a[0], a[1], b[0], b[1]] for a, b in pairs], dtype=np.int64).reshape(-1, 4)


def test_schedules_and_loop_test_bit_exact():
    gold = load_golden('schedules')
    n = 0
    for spec in C.schedule_topologies():
        g = O.Graph(spec)
        for vid in g.var_order:
            assert bool(gold['%s/loops_root%d' % (spec['name'], vid)]) == O.has_loops(g, vid)
            np.testing.assert_array_equal(gold['%s/sched_root%d' % (spec['name'], vid)],
                                          _sched_array(O.message_schedule(g, vid)))
            n += 1
    assert n == len([k for k in gold.files if '/sched_' in k])


def _run_case(case, approx=False):
    spec = case['spec']
    gold = load_golden(case['name'])
    inputs = C.make_inputs(spec, case['seed'], case['kind'] or 'uniform')
    g = O.Graph(spec)
    keys = C.msg_keys(spec)
    loopy = O.has_loops(g, case['roots'][0])
    assert loopy == bool(gold['is_loopy'])
    msgs = O.init_messages(g)
    assert sorted(msgs) == sorted(keys)
    np.testing.assert_array_equal(gold['msgs_init'], np.stack([msgs[k] for k in keys]))
    if case['force_loopy']:
        loopy = True
    if 'request' in case:
        ran = O.treelike_inference(g, inputs, msgs, case['request'], case['roots'], loopy, approx)
        assert ran == int(gold['roots_consumed'])
        np.testing.assert_allclose(np.stack([msgs[k] for k in keys]),
                                   gold['msgs_s%d' % case['snaps'][0]], rtol=RTOL, atol=0)
    else:
        done = 0
        for s in case['snaps']:
            while done < s:
                O.sweep(g, inputs, msgs, case['roots'][done], approx)
                done += 1
            np.testing.assert_allclose(np.stack([msgs[k] for k in keys]), gold['msgs_s%d' % s],
                                       rtol=RTOL, atol=1e-300)
    for r in sorted(set(case['roots'])):
        np.testing.assert_array_equal(gold['sched_root%d' % r], _sched_array(O.message_schedule(g, r)))
    np.testing.assert_array_equal(gold['var_order'], np.array(g.var_order))
    marg = np.stack([O.marginal(g, msgs, v) for v in g.var_order])
    np.testing.assert_allclose(marg, gold['marginals'], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(O.log_posterior(g, msgs), float(gold['log_posterior']), rtol=1e-12)
    if spec['X'] >= 50:
        top = np.stack([O.top_indices(O.marginal(g, msgs, v), 50) for v in g.var_order])
        np.testing.assert_array_equal(top, gold['top50'])
        np.testing.assert_array_equal(np.array(O.precision_counts(g, msgs)), gold['precision_counts'])
    if case.get('light'):
        return
    for f in g.factors:
        b = O.factor_beliefs(g, inputs, msgs, f['id'], approx)
        np.testing.assert_allclose(b, gold['belief_F%d' % f['id']], rtol=1e-11, atol=1e-300)
    if spec['style'] == 'trainmp':
        for f in g.factors:
            np.testing.assert_allclose(O.factor_gradient(g, inputs, msgs, f['id'], approx),
                                       gold['grad_F%d' % f['id']], rtol=1e-9, atol=1e-13)
        reg, lr = 0.2 / 17.0, 0.05     # set by make_golden.run_inference_case
        ee, ed = O.unregularized_gradient(g, inputs, msgs, approx)
        np.testing.assert_allclose(ee, gold['grad_unreg_en_en'], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(ed, gold['grad_unreg_en_de'], rtol=1e-9, atol=1e-13)
        ed2, ee2 = O.regularized_gradient(g, inputs, msgs, reg, approx)
        np.testing.assert_allclose(ee2, gold['grad_reg_en_en'], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(ed2, gold['grad_reg_en_de'], rtol=1e-9, atol=1e-13)
        ree, red = O.return_gradient(g, inputs, msgs, reg, lr, approx)
        np.testing.assert_allclose(ree, gold['grad_ret_en_en'], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(red, gold['grad_ret_en_de'], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('case', C.inference_cases(), ids=lambda c: c['name'])
def test_inference_case(case):
    _run_case(case)


@pytest.mark.parametrize('case', C.approx_cases(), ids=lambda c: c['name'])
def test_approx_case(case):
    _run_case(case, approx=True)


def test_tree_marginals_match_brute_force():
    """Independent known-answer test: on a tree one sweep is exact, so marginals must equal
    enumeration of the joint."""
    import itertools
    spec = C.chain_spec(4, 3)
    inputs = C.make_inputs(spec, 3, 'uniform')
    g, msgs, ran = O.run(spec, inputs, [2], 5)
    assert ran == 1
    X, n = 3, 4
    joint = np.zeros((X,) * n)
    for xs in itertools.product(range(X), repeat=n):
        p = 1.0
        for f in spec['factors']:
            T = inputs['tables'][f['table']]
            p *= T[xs[f['vars'][0]], 0] if len(f['vars']) == 1 else T[xs[f['vars'][0]], xs[f['vars'][1]]]
        joint[xs] = p
    joint /= joint.sum()
    for v in range(n):
        ax = tuple(a for a in range(n) if a != v)
        np.testing.assert_allclose(O.marginal(g, msgs, v), joint.sum(axis=ax), rtol=1e-12)


# ---------------------------------------------------------------------------------------------
# array primitives
# ---------------------------------------------------------------------------------------------
def _au_inputs(X, seed):
    rs = np.random.RandomState(seed)
    return dict(m1=rs.rand(X, 1), m2=rs.rand(X, 1), T=rs.rand(X, X) + 0.01, T2=np.exp(rs.randn(X, X)),
                c=rs.rand(X, 1) ** 4, r=rs.rand(1, X) ** 4)


AU_SIZES = (('x4', 4, 4004), ('x64', 64, 4064), ('x128', 128, 4128), ('x128b', 128, 5128))


@pytest.mark.parametrize('tag,X,seed', AU_SIZES)
def test_au_functions(tag, X, seed):
    gold = load_golden('au_functions')
    i = _au_inputs(X, seed)
    p = tag + '/'
    eq = lambda a, k: np.testing.assert_allclose(a, gold[p + k], rtol=1e-13, atol=0)  # noqa: E731
    eq(au.pointwise_multiply(i['m1'], i['m2']), 'pointwise_multiply')
    eq(au.dense_pointwise_multiply(i['T'], i['T2']), 'dense_pointwise_multiply')
    eq(au.normalize(i['m1'].copy()), 'normalize_vec')
    eq(au.normalize(i['T'].copy()), 'normalize_mat')
    z = np.zeros((X, 1))
    assert (au.normalize(z) is z) == bool(gold[p + 'normalize_zero_is_same_object'])
    neg = -i['m1']
    eq(au.normalize(neg), 'normalize_negative_sum')
    eq(neg, 'normalize_negative_sum_inplace')
    eq(au.dense_dot(i['T'], i['m1']), 'dense_dot_Tm')
    eq(au.dense_dot(i['m1'].T, i['T']), 'dense_dot_mT')
    eq(au.dense_dot(i['c'], i['r']), 'dense_dot_outer')
    assert au.dense_dot(i['m1'].T, i['T']).shape == (1, X)
    if X >= 100:
        col = au.sparse_vec_mat_dot(i['c'], i['T'])
        row = au.sparse_vec_mat_dot(i['r'], i['T'])
        assert col.shape == (X, 1) and row.shape == (X,)
        eq(col, 'sparse_vec_mat_dot_col')
        eq(row, 'sparse_vec_mat_dot_row')
        sp, ci, ri = au.sparse_dot(i['c'], i['r'])
        eq(sp, 'sparse_dot')
        np.testing.assert_array_equal(np.sort(ci), gold[p + 'sparse_dot_cidx_sorted'])
        np.testing.assert_array_equal(np.sort(ri), gold[p + 'sparse_dot_ridx_sorted'])
        spm = au.sparse_pointwise_multiply(sp, ci, ri, i['T'])
        eq(spm, 'sparse_pointwise_multiply')
        spn = au.sparse_normalize(spm, ci, ri)
        assert spn is spm
        eq(spn, 'sparse_normalize')


def test_au_error_behaviour_matches_recorded_reference_errors():
    errs = json.load(open(os.path.join(GOLDEN, 'MANIFEST.json')))['au_errors']

    def rec(fn):
        try:
            fn()
        except BaseException as e:  # noqa: B902
            return '%s: %s' % (type(e).__name__, e)
        return 'no exception'
    i = _au_inputs(8, 1)
    assert rec(lambda: au.dense_dot(i['T'].astype(np.float32), i['m1'])).split(',')[0] == \
        errs['dense_dot_float32'].split(',')[0]
    assert rec(lambda: au.dense_dot(i['T'], i['m1'].reshape(-1))) == errs['dense_dot_ndim1']
    for tag, X, seed in AU_SIZES[:2]:
        j = _au_inputs(X, seed)
        assert rec(lambda: au.sparse_vec_mat_dot(j['c'], j['T'])) == errs[tag + '/sparse_vec_mat_dot_col']
        assert rec(lambda: au.sparse_dot(j['c'], j['r'])) == errs[tag + '/sparse_dot']
