"""The drop-in object API (macaronicusermodeling_amd.LBP) driven through exactly the calls the
generator made on the reference, compared with the reference's recorded outputs.  These tests read
like the reference's own usage: build_graph() is shared with tests/golden/make_golden.py."""
import types

import numpy as np
import pytest

import cases as C
from conftest import load_golden

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

RTOL = 1e-10


class Roots:
    def __init__(self, L):
        self.queue = []
        L.random = types.SimpleNamespace(sample=self.sample)

    def sample(self, population, k):
        r = self.queue.pop(0)
        assert r in population
        return [r]


@pytest.fixture()
def L():
    import importlib
    import macaronicusermodeling_amd.LBP as mod
    mod = importlib.reload(mod)
    return mod


def _stack(fg, keys):
    return np.stack([fg.messages[k].m.reshape(-1) for k in keys])


def _run(L, case, approx=False):
    spec = case['spec']
    gold = load_golden(case['name'])
    inputs = C.make_inputs(spec, case['seed'], case['kind'] or 'uniform')
    roots = Roots(L)
    fg = C.build_graph(L, spec, inputs)
    fg.learning_rate = 0.05
    fg.regularization_param = 0.2 / 17.0
    if approx:
        fg.use_approx_inference = True
        fg.use_approx_beliefs = True
    keys = C.msg_keys(spec)
    roots.queue = [case['roots'][0]]
    fg.initialize()
    assert bool(fg.isLoopy) == bool(gold['is_loopy'])
    assert sorted(fg.messages.keys()) == sorted(keys)
    np.testing.assert_array_equal(_stack(fg, keys), gold['msgs_init'])
    assert fg.messages[keys[0]].m.shape == (spec['X'], 1)
    if case['force_loopy']:
        fg.isLoopy = True
    if 'request' in case:
        roots.queue = list(case['roots'])
        fg.treelike_inference(case['request'])
        assert len(case['roots']) - len(roots.queue) == int(gold['roots_consumed'])
        np.testing.assert_allclose(_stack(fg, keys), gold['msgs_s%d' % case['snaps'][0]], rtol=RTOL, atol=1e-300)
    else:
        done = 0
        for s in case['snaps']:
            roots.queue = list(case['roots'][done:s])
            fg.treelike_inference(s - done)
            done = s
            np.testing.assert_allclose(_stack(fg, keys), gold['msgs_s%d' % s], rtol=RTOL, atol=1e-300)
    for r in sorted(set(case['roots'])):
        sched = fg.get_message_schedule(fg.variables[r])
        enc = np.array([[*C.node_code(str(a)), *C.node_code(str(b))] for a, b in sched], dtype=np.int64)
        np.testing.assert_array_equal(enc, gold['sched_root%d' % r])
    vorder = list(fg.variables.keys())
    np.testing.assert_array_equal(np.array(vorder), gold['var_order'])
    marg = np.stack([fg.variables[v].get_marginal().m.reshape(-1) for v in vorder])
    np.testing.assert_allclose(marg, gold['marginals'], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(fg.get_posterior_probs(), float(gold['log_posterior']), rtol=1e-10)
    if spec['X'] >= 50:
        top = [[int(w[1:]) for w, _ in fg.variables[v].get_max_vocab(50)[2]] for v in vorder]
        np.testing.assert_array_equal(np.array(top), gold['top50'])
        np.testing.assert_array_equal(np.array(fg.get_precision_counts()), gold['precision_counts'])
    if case.get('light'):
        return fg
    for f in fg.factors:
        np.testing.assert_allclose(f.get_factor_beliefs(), gold['belief_F%d' % f.id], rtol=1e-10, atol=1e-300)
    if spec['style'] == 'trainmp':
        for f in fg.factors:
            g = f.get_gradient()
            assert g.shape == gold['grad_F%d' % f.id].shape
            np.testing.assert_allclose(g, gold['grad_F%d' % f.id], rtol=1e-8, atol=1e-12)
        ee, ed = fg.get_unregularized_gradeint()
        np.testing.assert_allclose(ee, gold['grad_unreg_en_en'], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(ed, gold['grad_unreg_en_de'], rtol=1e-8, atol=1e-12)
        ed2, ee2 = fg.get_gradient()
        np.testing.assert_allclose(ee2, gold['grad_reg_en_en'], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(ed2, gold['grad_reg_en_de'], rtol=1e-8, atol=1e-12)
        ree, red = fg.return_gradient()
        np.testing.assert_allclose(ree, gold['grad_ret_en_en'], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(red, gold['grad_ret_en_de'], rtol=1e-8, atol=1e-12)
    return fg


@pytest.mark.parametrize('case', C.inference_cases(), ids=lambda c: c['name'])
def test_object_api_matches_reference(L, case):
    _run(L, case)


@pytest.mark.parametrize('case', C.approx_cases(), ids=lambda c: c['name'])
def test_object_api_approximate_modes(L, case):
    _run(L, case, approx=True)


def test_single_message_updates_and_message_assignment(L):
    """VariableNode / FactorNode.update_message_to one at a time (the loop of LBP.py:227-243 written
    out by hand) must equal the fused sweep; writing graph.messages[...] must reach the device."""
    case = [c for c in C.inference_cases() if c['name'] == 'shuffled_x64'][0]
    spec = case['spec']
    inputs = C.make_inputs(spec, case['seed'], case['kind'])
    roots = Roots(L)
    fg1 = C.build_graph(L, spec, inputs)
    fg2 = C.build_graph(L, spec, inputs)
    for fg in (fg1, fg2):
        roots.queue = [2]
        fg.initialize()
    roots.queue = [7]
    fg1.treelike_inference(1)
    sched = fg2.get_message_schedule(fg2.variables[7])
    for frm, to in reversed(sched):
        if not (isinstance(to, L.FactorNode) and len(to.varset) < 2):
            frm.update_message_to(to)
    for to, frm in sched:
        if not (isinstance(to, L.FactorNode) and len(to.varset) < 2):
            frm.update_message_to(to)
    keys = C.msg_keys(spec)
    np.testing.assert_allclose(_stack(fg2, keys), _stack(fg1, keys), rtol=1e-13, atol=0)
    # assignment through the dict view
    k = keys[3]
    m = L.Message.new_message(C.domain_of(64), 0.0)
    m.m[5, 0] = 1.0
    fg2.messages[k] = m
    assert fg2.messages[k].m[5, 0] == 1.0 and fg2.messages[k].m.sum() == 1.0
    with pytest.raises(KeyError):
        fg2.messages['X_999', 'F_0']


def test_module_level_helpers(L):
    a = L.Message(np.array([0.2, 0.0, 0.6, 0.2]))
    b = L.Message(np.array([[0.5], [np.inf], [0.5], [0.0]]))
    with np.errstate(all='ignore'):
        want = np.nan_to_num(a.m * b.m)
    np.testing.assert_array_equal(L.pointwise_multiply(a, b).m, want)
    z = L.Message(np.zeros(8))
    z.renormalize()
    np.testing.assert_array_equal(z.m, np.full((8, 1), 0.125))
    m = L.Message(np.array([1.0, 3.0]))
    m.renormalize()
    np.testing.assert_allclose(m.m.reshape(-1), [0.25, 0.75], rtol=1e-15)
    assert L.VAR_TYPE_PREDICTED == 'var_type_predicted' and L.UNARY_FACTOR == 'unary_factor'
    with pytest.raises(NotImplementedError):
        f = L.FactorNode(0)
        vs = [L.VariableNode(i, L.VAR_TYPE_PREDICTED, 'en', ['a', 'b'], 'a') for i in range(3)]
        f.add_varset_with_potentials(vs, L.PotentialTable({0: 0, 1: 1, 2: 2}, table=np.ones((2, 2))))


def test_to_string_and_to_dist_formats(L):
    """Wire format of the prediction / .dist text (LBP.py:109-143), checked on a user graph."""
    case = [c for c in C.inference_cases() if c['name'] == 'user_k3_x64'][0]
    spec = case['spec']
    inputs = C.make_inputs(spec, case['seed'])
    roots = Roots(L)
    fg = C.build_graph(L, spec, inputs)
    for f in fg.factors:
        f.word_label = 'w%d' % (f.position or 0)
    roots.queue = [1]
    fg.initialize()
    roots.queue = [1, 4, 7]
    fg.treelike_inference(3)
    gold = load_golden(case['name'])
    lines = fg.to_dist().split('\n')
    assert len(lines) == 3
    truth, guess, logs = lines[0].split(' ||| ')
    assert truth == 'None' and guess == 'w%d' % spec['labels'][0]
    vals = np.array([float(x) for x in logs.split()])
    np.testing.assert_allclose(vals, np.log(gold['marginals'][0]), atol=6e-7)
    strings = fg.to_string()
    assert len(strings) == 10            # one line per sentence position (3 predicted + 7 given)
    toks = [s for s in strings if s.startswith('w1 ')][0].split()
    assert len(toks) == 3 + 2 * 50


def test_prediction_and_dist_text_equal_the_reference_output(L):
    """SURVEY 8 row f4, pinned: tests/golden/user_k3_x64_text.json holds the text the reference's own to_string /
    to_dist (LBP.py:109-143) and the '*SENT_ID:' block (train_mp.py:337) produce for the user_k3_x64 case after three
    sweeps (make_text_golden.py runs the reference).  The drop-in must emit the same characters: labels, the top-50
    vocabulary in the same order, '%0.4f' / '%0.6f' log-marginals."""
    import json
    import os
    from macaronicusermodeling_amd import tidir
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'user_k3_x64_text.json'), encoding='utf8'))
    case = [c for c in C.inference_cases() if c['name'] == gold['case']][0]
    spec = case['spec']
    roots = Roots(L)
    fg = C.build_graph(L, spec, C.make_inputs(spec, case['seed']))
    for f in fg.factors:
        f.word_label = 'w%d' % (f.position or 0)
    roots.queue = [gold['roots'][0]]
    fg.initialize()
    roots.queue = list(gold['roots'])
    fg.treelike_inference(3)
    assert fg.to_string() == gold['to_string']
    assert fg.to_dist() == gold['to_dist']
    assert tidir.prediction_block(17, fg) == gold['prediction_block']


def test_graphs_of_one_shape_share_their_programs_and_messages_are_copies(L):
    """Two FactorGraph objects of the same shape reuse one device program per root sequence (per-thread cache in
    batch.py), and `messages[k]` hands out copies: an in-place edit is not seen, an assignment is."""
    case = [c for c in C.inference_cases() if c['name'] == 'user_k3_x64'][0]
    spec = case['spec']
    roots = Roots(L)
    graphs = []
    for seed in (case['seed'], case['seed'] + 1):
        fg = C.build_graph(L, spec, C.make_inputs(spec, seed))
        roots.queue = [1]
        fg.initialize()
        roots.queue = [1, 4, 7]
        fg.treelike_inference(3)
        graphs.append(fg)
    e0, e1 = graphs[0]._engine, graphs[1]._engine
    assert e0.batch.program((1, 4, 7)) is e1.batch.program((1, 4, 7))
    assert not np.array_equal(_stack(graphs[0], e0.keys), _stack(graphs[1], e1.keys))      # different tables, own messages
    k = e0.keys[3]
    m = graphs[0].messages[k]
    before = m.m.copy()
    m.m[:] = 0.5                                         # a copy: nothing changes on the graph
    np.testing.assert_array_equal(graphs[0].messages[k].m, before)
    m2 = L.Message(np.full_like(before, 0.25))
    graphs[0].messages[k] = m2
    np.testing.assert_array_equal(graphs[0].messages[k].m.reshape(-1), np.full(before.size, 0.25))


def test_update_theta_and_the_toy_test_flow(L):
    """BASELINE config 1's flow through the drop-in (toy_test.py:143-154 with today's constructor): per instance
    initialize -> treelike_inference(3) -> get_gradient -> theta += 0.05 grad, the thetas carried to the next instance; and
    FactorGraph.update_theta (LBP.py:329-333), which does the same step with the graph's own learning rate, in place.
    Both against the reference's fixtures (grad_reg_* / grad_ret_* of user_k3_x64, learning rate 0.05)."""
    case = [c for c in C.inference_cases() if c['name'] == 'user_k3_x64'][0]
    spec, gold = case['spec'], load_golden(case['name'])
    inputs = C.make_inputs(spec, case['seed'], case['kind'] or 'uniform')
    theta_ee, theta_ed = inputs['theta_en_en'].copy(), inputs['theta_en_de'].copy()
    roots = Roots(L)
    for instance in range(2):                              # two instances of one shape, theta carried over (toy_test.py:151-152)
        ins = dict(inputs, theta_en_en=theta_ee, theta_en_de=theta_ed)
        fg = C.build_graph(L, spec, ins)
        fg.learning_rate, fg.regularization_param = 0.05, 0.2 / 17.0
        roots.queue = [case['roots'][0]]
        fg.initialize()
        roots.queue = list(case['roots'][:3])
        fg.treelike_inference(3)
        before_ee, before_ed = fg.theta_en_en.copy(), fg.theta_en_de.copy()
        grad_en_de, grad_en_en = fg.get_gradient()
        if instance == 0:
            np.testing.assert_allclose(grad_en_en, gold['grad_reg_en_en'], rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(grad_en_de, gold['grad_reg_en_de'], rtol=1e-8, atol=1e-12)
        fg.theta_en_en += 0.05 * grad_en_en
        fg.theta_en_de += 0.05 * grad_en_de
        np.testing.assert_allclose(fg.theta_en_en, before_ee + 0.05 * grad_en_en, rtol=0, atol=0)
        theta_ee, theta_ed = fg.theta_en_en, fg.theta_en_de
    # update_theta: the same step through the method, on the arrays the graph holds (callers keep references)
    fg = C.build_graph(L, spec, inputs)
    fg.learning_rate, fg.regularization_param = 0.05, 0.2 / 17.0
    roots.queue = [case['roots'][0]]
    fg.initialize()
    roots.queue = list(case['roots'][:3])
    fg.treelike_inference(3)
    held_ee, held_ed = fg.theta_en_en, fg.theta_en_de
    start_ee, start_ed = held_ee.copy(), held_ed.copy()
    out_ee, out_ed = fg.update_theta()
    assert out_ee is held_ee and out_ed is held_ed and fg.theta_en_en is held_ee
    np.testing.assert_allclose(held_ee - start_ee, gold['grad_ret_en_en'], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(held_ed - start_ed, gold['grad_ret_en_de'], rtol=1e-7, atol=1e-12)
