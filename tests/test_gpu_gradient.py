"""Batched beliefs / gradient kernels and the train_mp-shaped step against reference fixtures and
the CPU oracle."""
import os

import numpy as np
import pytest

import cases as C
from conftest import load_golden
from helpers import batch_tables, case_inputs
from oracle import lbp_oracle as O

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

USER_CASES = [c for c in C.inference_cases() if c['spec']['style'] == 'trainmp']


def _meta(spec, topo):
    by_id = {f['id']: f for f in spec['factors']}
    pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
    kinds, obs = [], []
    for j in topo.unary_factors:
        f = by_id[topo.factor_ids[j]]
        kinds.append(2 if f['factor_type'] == 'en_de' else (0 if f['gap'] > 1 else 1))
        obs.append(f['observed_dim'])
    label_of = dict(zip(spec['var_ids'], spec['labels']))
    labels = [label_of[v] for v in topo.var_ids]
    return pair_phi, kinds, obs, labels


@pytest.mark.parametrize('case', USER_CASES, ids=lambda c: c['name'])
def test_batched_gradient_and_beliefs(case):
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = case['spec']
    gold = load_golden(case['name'])
    topo = GraphTopology.from_spec(spec)
    nb = 3
    inputs = case_inputs(case, nb)
    # all graphs of the batch share graph 0's feature tensors (they are batch-level inputs); their
    # potentials differ, which is what the kernels read per graph
    for i in inputs[1:]:
        for k in ('phi_en_en', 'phi_en_en_w1', 'phi_en_de'):
            i[k] = inputs[0][k]
    pair, unary = batch_tables(spec, topo, inputs)
    fb = FactorGraphBatch(topo, spec['X'], nb)
    if topo.P:
        fb.set_pair_tables(pair)
    fb.set_unary_tables(unary)
    pair_phi, kinds, obs, labels = _meta(spec, topo)
    fb.set_features(inputs[0]['phi_en_en'], inputs[0]['phi_en_en_w1'], inputs[0]['phi_en_de'], pair_phi, kinds)
    fb.set_observations(np.tile(labels, (nb, 1)), np.tile(obs, (nb, 1)))
    fb.initialize(case['roots'][0])
    n = fb.treelike_inference(case.get('request', max(case['snaps'])), case['roots'] * 10)
    g_ee, g_ed = fb.gradient()
    assert _ffi.lib.mlbp_gradient_status() == 0
    g_ee, g_ed = g_ee.cpu().numpy(), g_ed.cpu().numpy()
    # the same gradients produced inside the sweep launch (fused when tables are register-resident)
    f_ee = torch.full((nb, 3), float('nan'), dtype=torch.float64, device=fb.device)
    f_ed = torch.full((nb, 6), float('nan'), dtype=torch.float64, device=fb.device)
    fb.initialize(case['roots'][0])
    prog = fb.sweep((case['roots'] * 10)[:n], gradient=(f_ee, f_ed))
    assert prog.status() == 0
    if spec['X'] == 64 and 1 <= topo.P <= 3:        # the default kernel carries the gradient as its epilogue
        assert _ffi.lib.mlbp_last_sweep_kernel() == 7 and prog.exact_count(nb) == 0
    np.testing.assert_allclose(f_ee.cpu().numpy(), g_ee, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(f_ed.cpu().numpy(), g_ed, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g_ee[0], gold['grad_unreg_en_en'].reshape(-1), rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(g_ed[0], gold['grad_unreg_en_de'].reshape(-1), rtol=1e-8, atol=1e-11)
    g = O.Graph(spec)
    for b in range(1, nb):
        msgs = O.init_messages(g)
        O.treelike_inference(g, inputs[b], msgs, n, case['roots'] * 10, True)
        ee, ed = O.unregularized_gradient(g, inputs[b], msgs)
        np.testing.assert_allclose(g_ee[b], ee.reshape(-1), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(g_ed[b], ed.reshape(-1), rtol=1e-8, atol=1e-11)
    if topo.P:
        bel = fb.pair_beliefs().cpu().numpy()
        for p, j in enumerate(topo.pair_factors):
            np.testing.assert_allclose(bel[0, p], gold['belief_F%d' % topo.factor_ids[j]], rtol=1e-10, atol=1e-300)
    tot = fb.sum_rows(torch.from_numpy(g_ed).to(fb.device)).cpu().numpy()
    np.testing.assert_allclose(tot, g_ed.sum(0), rtol=1e-13)


def test_zero_table_gives_zero_beliefs_like_au_normalize():
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    case = USER_CASES[0]
    spec = case['spec']
    topo = GraphTopology.from_spec(spec)
    inputs = case_inputs(case, 1)
    pair, unary = batch_tables(spec, topo, inputs)
    pair[0] = 0.0
    fb = FactorGraphBatch(topo, spec['X'], 1)
    fb.set_pair_tables(pair); fb.set_unary_tables(unary)
    fb.initialize()
    assert float(fb.pair_beliefs()[0, 0].abs().sum()) == 0.0


def _instances(spec, topo, B, seed):
    """Per-instance labels and observed columns (what differs between train_mp instances of one
    sentence shape)."""
    rs = np.random.RandomState(seed)
    by_id = {f['id']: f for f in spec['factors']}
    X, Vde = spec['X'], spec['Vde']
    labels = rs.randint(0, X, size=(B, topo.n_vars))
    obs = np.zeros((B, topo.U), dtype=np.int64)
    for u, j in enumerate(topo.unary_factors):
        f = by_id[topo.factor_ids[j]]
        obs[:, u] = rs.randint(0, Vde if f['factor_type'] == 'en_de' else X, size=B)
    return labels, obs


def _oracle_step(spec, inputs, labels, obs, roots, lr, reg):
    """sum over instances of return_gradient + log posterior, by the oracle."""
    import copy
    tot_ee = np.zeros_like(inputs['theta_en_en']); tot_ed = np.zeros_like(inputs['theta_en_de']); lp = 0.0
    topo_vars = None
    for b in range(labels.shape[0]):
        s = copy.deepcopy(spec)
        g0 = O.Graph(s)
        topo_vars = g0.var_order
        s['labels'] = [int(labels[b, topo_vars.index(v)]) for v in s['var_ids']]
        unary_ids = [f['id'] for f in g0.factors if len(f['vars']) == 1]
        for f in s['factors']:
            if len(f['vars']) == 1:
                f['observed_dim'] = int(obs[b, unary_ids.index(f['id'])])
        g = O.Graph(s)
        msgs = O.init_messages(g)
        O.treelike_inference(g, inputs, msgs, len(roots), roots, O.has_loops(g, roots[0]))
        ee, ed = O.return_gradient(g, inputs, msgs, reg, lr)
        tot_ee += ee; tot_ed += ed
        lp += O.log_posterior(g, msgs)
    return tot_ee, tot_ed, lp


@pytest.mark.parametrize('planes', ['random', 'reference', 'k4', 'k5'])
@pytest.mark.parametrize('B', [6, 53])
def test_shared_table_sweep_runs_the_gradient_as_its_epilogue(B, planes):
    """Shared pots, K3 user graphs: the gradient of sweep(gradient=...) comes out of the shared-table sweep kernel itself (the
    final variable->factor messages of the workgroup's 16 graphs against T (.) phi_k on the matrix cores, the unary part by
    gather in the launch in front of it) -- the separate gradient launch's values to rounding, the oracle's values, and a graph
    the matrix-core kernel hands to the exact kernel (a zero table column) gets its gradient there.  planes = 'reference': the
    zero plane and the bias plane of the reference's tensors, which the epilogue recognises and does not contract.
    'k4': four predicted words (train_mp.py:272-282 builds the complete graph over them: 6 pairwise factors, three-source
    variable updates, 13 of 21 message tiles spilled) -- the same epilogue in the kernel's general instance; a flagged graph
    gets its gradient from the per-graph kernel run on the flagged graphs only.  'k5': ten pairwise factors, four-source updates,
    most of the message tiles in memory -- the general instance again (the per-graph kernels are 4 x slower on shared tables)."""
    import copy
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    pred = {'random': [1, 4, 7], 'reference': [1, 2, 7], 'k4': [1, 2, 5, 8], 'k5': [0, 1, 4, 6, 9]}[planes]
    spec = C.user_spec(10, pred, 64, 48, seed=1)     # ('reference', 'k4': both pots in use)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 77)
    if planes == 'reference':
        inputs = C.reference_planes(inputs)
    roots = list(spec['var_ids'])[1:] + list(spec['var_ids'])[:1]
    labels, obs = _instances(spec, topo, B, 5)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=roots)
    fb = tr.batch
    tr.build_potentials()
    g_ee, g_ed = torch.full_like(tr._g_ee, float('nan')), torch.full_like(tr._g_ed, float('nan'))
    fb.sweep(roots, init=True, marginals=tr._marg, gradient=(g_ee, g_ed), keep_messages=False)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3 and _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    assert fb.program(roots).exact_count(B) == 0
    fb.sweep(roots, init=True)                                    # all messages to memory, then the separate launch
    assert _ffi.lib.mlbp_last_sweep_fused_gradient() == 0
    h_ee, h_ed = fb.gradient()
    # (the same pairwise operations in the same order; the unary terms are summed per graph in the prepare launch, in another order)
    np.testing.assert_allclose(g_ee.cpu().numpy(), h_ee.cpu().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(g_ed.cpu().numpy(), h_ed.cpu().numpy(), rtol=1e-12, atol=1e-13)
    for b in range(0, B, 7):
        s = copy.deepcopy(spec)
        g0 = O.Graph(s)
        s['labels'] = [int(labels[b, g0.var_order.index(v)]) for v in s['var_ids']]
        unary_ids = [f['id'] for f in g0.factors if len(f['vars']) == 1]
        for f in s['factors']:
            if len(f['vars']) == 1:
                f['observed_dim'] = int(obs[b, unary_ids.index(f['id'])])
        g = O.Graph(s)
        msgs = O.init_messages(g)
        O.treelike_inference(g, inputs, msgs, len(roots), roots, O.has_loops(g, roots[0]))
        ee, ed = O.unregularized_gradient(g, inputs, msgs)
        np.testing.assert_allclose(g_ee[b].cpu().numpy(), ee.reshape(-1), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g_ed[b].cpu().numpy(), ed.reshape(-1), rtol=1e-9, atol=1e-12)
    # one graph reads an all-zero unary row: flagged by the prepare launch, redone (gradient included) by the exact kernel
    row = int(fb.unary_tab[B // 2, 0])
    saved = tr.unary_tables[row].clone()
    tr.unary_tables[row] = 0.0
    fb._uexp_rows_done = 0          # (the table changes behind the trainer's back: the rows' expected features come from the tables again, not from the potentials launch)
    k_ee, k_ed = torch.full_like(g_ee, float('nan')), torch.full_like(g_ed, float('nan'))
    fb.sweep(roots, init=True, marginals=tr._marg, gradient=(k_ee, k_ed), keep_messages=False)
    n_redone = fb.program(roots).exact_count(B)
    assert _ffi.lib.mlbp_last_sweep_fused_gradient() == 1 and n_redone >= 1
    try:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))
        e_ee, e_ed = torch.empty_like(g_ee), torch.empty_like(g_ed)
        fb.sweep(roots, init=True, marginals=tr._marg, gradient=(e_ee, e_ed), keep_messages=False)
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    assert bool(torch.isfinite(k_ee).all()) and bool(torch.isfinite(k_ed).all())
    np.testing.assert_allclose(k_ee.cpu().numpy(), e_ee.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(k_ed.cpu().numpy(), e_ed.cpu().numpy(), rtol=1e-9, atol=1e-12)
    tr.unary_tables[row] = saved


def test_fused_gradient_with_and_without_the_message_write_back():
    """K3 in the product-fused form: the gradient epilogue reads the final variable->factor messages from the kernel's tiles, so
    it is the same bits whether the call also writes the messages back (the members then stash their raw results and store
    the slots' last values) or not; and the messages written beside the gradient are those of a plain sweep."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(10, [1, 2, 7], 64, 48, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = C.reference_planes(C.make_inputs(spec, 78))
    roots = list(spec['var_ids'])
    assert topo.plan(roots)['shared_gradient_from_tiles'] == 1
    B = 41
    labels, obs = _instances(spec, topo, B, 6)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=roots)
    fb = tr.batch
    tr.build_potentials()
    a_ee, a_ed = torch.full_like(tr._g_ee, float('nan')), torch.full_like(tr._g_ed, float('nan'))
    m_a = torch.full_like(tr._marg, float('nan'))
    fb.sweep(roots, init=True, marginals=m_a, gradient=(a_ee, a_ed), keep_messages=False)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3 and _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    b_ee, b_ed = torch.full_like(a_ee, float('nan')), torch.full_like(a_ed, float('nan'))
    m_b = torch.full_like(m_a, float('nan'))
    fb.msgs.fill_(float('nan'))
    fb.sweep(roots, init=True, marginals=m_b, gradient=(b_ee, b_ed), keep_messages=True)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3 and _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    assert torch.equal(a_ee, b_ee) and torch.equal(a_ed, b_ed) and torch.equal(m_a, m_b)
    with_gradient = fb.msgs.clone()
    assert bool(torch.isfinite(with_gradient).all())
    fb.msgs.fill_(float('nan'))
    fb.sweep(roots, init=True)
    torch.testing.assert_close(with_gradient, fb.msgs, rtol=1e-12, atol=1e-300)


def test_train_step_matches_sum_of_reference_steps():
    """UserGraphTrainer.step == theta + sum_i return_gradient_i (train_mp.py:398, 419-424) with the
    potentials built on the device from phi and theta (train_mp.py:220-255)."""
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(10, [1, 4, 7], 64, 48, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 77)
    B, roots, lr, reg = 6, [4, 1, 7], 0.1, 0.2 / 6
    labels, obs = _instances(spec, topo, B, 5)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=roots)
    tr.build_potentials()
    np.testing.assert_allclose(tr.pair_tables[0].cpu().numpy(), inputs['pot_en_en'], rtol=1e-14)
    np.testing.assert_allclose(tr.unary_tables[2 * 64:].cpu().numpy(), inputs['pot_en_de'].T, rtol=1e-14)
    mean_lp, t_ee, t_ed = tr.step(lr, reg)
    s_ee, s_ed, lp = _oracle_step(spec, inputs, labels, obs, roots, lr, reg)
    np.testing.assert_allclose(t_ee.cpu().numpy(), (inputs['theta_en_en'] + s_ee).reshape(-1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t_ed.cpu().numpy(), (inputs['theta_en_de'] + s_ed).reshape(-1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mean_lp, lp / B, rtol=1e-10)


def test_synthetic_ti_dir_bucket_through_the_trainer(tmp_path):
    """TI_DIR files -> parsed instances -> shape buckets -> UserGraphTrainer statistics, against the
    oracle run per instance on the spec that instance implies."""
    import copy
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import UserGraphTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=60, X=64, Vde=64, sent_len=(5, 7), n_predicted=(3, 3), seed=5)
    en, de = tidir.read_vocab(paths['end']), tidir.read_vocab(paths['ded'])
    phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
    buckets = tidir.bucket_instances(tidir.read_instances(paths['ti']), en, de)
    key, b = max(buckets.items(), key=lambda kv: len(kv[1]['rows']))
    rs = np.random.RandomState(1)
    theta_ee, theta_ed = rs.randn(1, 3) * 0.3, rs.randn(1, 6) * 0.3
    roots = [key[1][0], key[1][1], key[1][2]]
    tr = UserGraphTrainer(b['spec'], b['var_labels'], b['unary_obs'], phi_ee, phi_w1, phi_ed, theta_ee, theta_ed, roots=roots)
    stats = tr.local_statistics().cpu().numpy()
    inputs = dict(phi_en_en=phi_ee, phi_en_en_w1=phi_w1, phi_en_de=phi_ed, theta_en_en=theta_ee, theta_en_de=theta_ed,
                  pot_en_en=np.exp(phi_ee.dot(theta_ee.T).reshape(64, 64)),
                  pot_en_en_w1=np.exp(phi_w1.dot(theta_ee.T).reshape(64, 64)),
                  pot_en_de=np.exp(phi_ed.dot(theta_ed.T).reshape(64, 64)))
    want = np.zeros(3 + 6 + 2)
    unary = [f for f in sorted(b['spec']['factors'], key=lambda f: f['id']) if len(f['vars']) == 1]
    for i in range(len(b['rows'])):
        s = copy.deepcopy(b['spec'])
        s['labels'] = [int(v) for v in b['var_labels'][i]]
        for u, f in enumerate(unary):
            s['factors'][f['id']]['observed_dim'] = int(b['unary_obs'][i, u])
        g = O.Graph(s)
        msgs = O.init_messages(g)
        O.treelike_inference(g, inputs, msgs, 3, roots, O.has_loops(g, roots[0]))
        ee, ed = O.unregularized_gradient(g, inputs, msgs)
        want[:3] += ee.reshape(-1); want[3:9] += ed.reshape(-1); want[9] += O.log_posterior(g, msgs); want[10] += 1
    np.testing.assert_allclose(stats, want, rtol=1e-8, atol=1e-10)


def test_tidir_trainer_epochs_and_predictions(tmp_path):
    """End to end over files: three epochs of TiDirTrainer (all shape buckets, shared theta) against
    the same loop driven by the oracle; params file written in the reference's format; prediction
    counts against the oracle's precision counts."""
    import copy
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=24, X=64, Vde=64, sent_len=(4, 6), n_predicted=(1, 3), seed=9)
    tt = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                      paths['phi_ped'], sweeps=3)
    assert sum(len(b['rows']) for b in tt.buckets.values()) == 24 and len(tt.buckets) > 1
    save = os.path.join(str(tmp_path), 'params')
    hist = tt.train(epochs=2, reg_param=0.2, save_params=save)
    # the oracle's version of the same two epochs
    phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
    th_ee, th_ed = np.zeros((1, 3)), np.zeros((1, 6))

    feat = {'correct': 2, 'full_history': 3, 'hit_history': 4}

    def oracle_pass(th_ee, th_ed, want_counts=False):
        pot_ee = np.exp(phi_ee.dot(th_ee.T).reshape(64, 64)); pot_w1 = np.exp(phi_w1.dot(th_ee.T).reshape(64, 64))
        tot = np.zeros(11); counts = np.zeros(4, dtype=np.int64); n_cells = 0
        for key, b in sorted(tt.buckets.items()):
            unary = [f for f in sorted(b['spec']['factors'], key=lambda f: f['id']) if len(f['vars']) == 1]
            roots = [key[1][i % len(key[1])] for i in range(3)]
            for i in range(len(b['rows'])):
                s = copy.deepcopy(b['spec'])
                s['labels'] = [int(v) for v in b['var_labels'][i]]
                for u, f in enumerate(unary):
                    s['factors'][f['id']]['observed_dim'] = int(b['unary_obs'][i, u])
                # the reference writes the three planes into phi_en_de for this instance (train_mp.py:178-217)
                phi_i = phi_ed.copy()
                for name, k in feat.items():
                    plane = np.zeros((64, 64))
                    for ci, cj, cv in b['rows'][i]['planes'][name]:
                        plane[ci, cj] += cv; n_cells += 1
                    phi_i[:, :, k] = plane
                inputs = dict(phi_en_en=phi_ee, phi_en_en_w1=phi_w1, phi_en_de=phi_i, theta_en_en=th_ee, theta_en_de=th_ed,
                              pot_en_en=pot_ee, pot_en_en_w1=pot_w1, pot_en_de=np.exp(phi_i.dot(th_ed.T).reshape(64, 64)))
                g = O.Graph(s); msgs = O.init_messages(g)
                O.treelike_inference(g, inputs, msgs, 3, roots, O.has_loops(g, roots[0]))
                ee, ed = O.unregularized_gradient(g, inputs, msgs)
                tot[:3] += ee.reshape(-1); tot[3:9] += ed.reshape(-1); tot[9] += O.log_posterior(g, msgs); tot[10] += 1
                if want_counts:
                    counts += np.array(O.precision_counts(g, msgs))
        assert n_cells > 0
        return tot, counts
    for epoch in range(2):
        lr, reg = 0.1 / (1 + 0.3 * epoch), 0.2 / 24
        tot, _ = oracle_pass(th_ee, th_ed)
        np.testing.assert_allclose(hist[epoch], tot[9] / 24, rtol=1e-9)
        th_ee = th_ee + lr * (tot[:3] - 24 * reg * th_ee)
        th_ed = th_ed + lr * (tot[3:9] - 24 * reg * th_ed)
    np.testing.assert_allclose(tt.theta_en_en.cpu().numpy(), th_ee.reshape(-1), rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(tt.theta_en_de.cpu().numpy(), th_ed.reshape(-1), rtol=1e-8, atol=1e-11)
    assert sum(tr.n_priv for tr in tt.trainers.values()) > 0 and abs(th_ed[0, 2]) > 0     # the planes were live
    een, eet, edn, edt, _ = tidir.read_params(save)
    np.testing.assert_allclose(eet.reshape(-1), th_ee.reshape(-1), atol=1e-6)
    assert os.path.exists(save + '.iter0') and os.path.exists(save + '.iter1')
    mean_lp, counts = tt.predict()
    tot, want_counts = oracle_pass(th_ee, th_ed, want_counts=True)
    np.testing.assert_allclose(mean_lp, tot[9] / 24, rtol=1e-9)
    assert counts == tuple(int(v) for v in want_counts)


@pytest.mark.parametrize('X', [128, 203, 777, 1100])
def test_train_step_at_a_large_state_space_uses_batched_gemms(X):
    """X = 128 / 203 (a vocabulary size that is no multiple of anything: zero-padded to 256) / 777 (the one-image kernel at 896)
    / 1100 (the chunked kernel, 2 x 2 blocks of 640): the trainer's shared pots make every factor->variable update of the shard
    one MFMA contraction over the batch, and every pairwise factor's share of the gradient one more (four weighted tables side by
    side, the dot with the other message in its epilogue); same step as the oracle's per-instance loop."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(8, [1, 3, 6], X, 40, seed=2)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 78)
    B, roots, lr, reg = 5, [3, 1, 6], 0.1, 0.2 / 5
    labels, obs = _instances(spec, topo, B, 6)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=roots)
    mean_lp, t_ee, t_ed = tr.step(lr, reg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 6 and tr.batch.program(roots).status() == 0
    s_ee, s_ed, lp = _oracle_step(spec, inputs, labels, obs, roots, lr, reg)
    np.testing.assert_allclose(t_ee.cpu().numpy(), (inputs['theta_en_en'] + s_ee).reshape(-1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t_ed.cpu().numpy(), (inputs['theta_en_de'] + s_ed).reshape(-1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mean_lp, lp / B, rtol=1e-10)
    # the pairwise part of that gradient ran as DGEMMs ((T (.) phi_k) . R, then a dot with c per graph): same numbers
    # from the per-graph kernel on the same messages
    fb = tr.batch
    fb.sweep(roots, init=True)
    ee1, ed1 = fb.gradient()
    fb.use_shared_gradient = False
    ee2, ed2 = fb.gradient()
    np.testing.assert_allclose(ee1.cpu().numpy(), ee2.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ed1.cpu().numpy(), ed2.cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert _ffi.lib.mlbp_gradient_status() == 0


def _pots(inputs, th_ee, th_ed, X=64):
    d = dict(inputs)
    d['theta_en_en'], d['theta_en_de'] = th_ee, th_ed
    d['pot_en_en'] = np.exp(inputs['phi_en_en'].dot(th_ee.T).reshape(X, X))
    d['pot_en_en_w1'] = np.exp(inputs['phi_en_en_w1'].dot(th_ee.T).reshape(X, X))
    d['pot_en_de'] = np.exp(inputs['phi_en_de'].dot(th_ed.T).reshape(X, -1))
    return d


def test_train_step_with_per_domain_thetas():
    """--user_adapt / --experience_adapt (train_mp.py:162-171, 226-247, 384-396, 413-415): instance i builds its
    potentials from theta_dom[d_i] instead of the global theta; the global theta takes every instance's step, each
    domain theta the steps of its own instances with the regulariser scaled.  Domain sizes 17 / 5 / 18 make groups
    of 16 graphs straddle domains: those groups leave the matrix-core kernels for the per-graph ones."""
    import copy
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(10, [1, 4, 7], 64, 48, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 77)
    rs = np.random.RandomState(3)
    D, sizes = 3, [17, 5, 18]
    B = sum(sizes)
    dom = np.repeat(np.arange(D), sizes)
    th_dom_ee, th_dom_ed = rs.randn(D, 3) * 0.3, rs.randn(D, 6) * 0.3
    roots, lr, reg, scale = [4, 1, 7], 0.1, 0.2 / B, 0.5
    labels, obs = _instances(spec, topo, B, 5)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=roots, domains=dom,
                          theta_dom_en_en=th_dom_ee.copy(), theta_dom_en_de=th_dom_ed.copy())
    stats = tr.local_statistics().cpu().numpy().copy()
    assert tr.batch.program(roots).exact_count(B) > 0          # the straddling groups were redone per graph
    want = np.zeros((1 + D, 11))
    g0 = O.Graph(copy.deepcopy(spec))
    unary_ids = [f['id'] for f in g0.factors if len(f['vars']) == 1]
    for b in range(B):
        s = copy.deepcopy(spec)
        s['labels'] = [int(labels[b, g0.var_order.index(v)]) for v in s['var_ids']]
        for f in s['factors']:
            if len(f['vars']) == 1:
                f['observed_dim'] = int(obs[b, unary_ids.index(f['id'])])
        inp = _pots(inputs, th_dom_ee[dom[b]].reshape(1, -1), th_dom_ed[dom[b]].reshape(1, -1))
        g = O.Graph(s); msgs = O.init_messages(g)
        O.treelike_inference(g, inp, msgs, 3, roots, O.has_loops(g, roots[0]))
        ee, ed = O.unregularized_gradient(g, inp, msgs)
        row = np.concatenate([ee.reshape(-1), ed.reshape(-1), [O.log_posterior(g, msgs), 1.0]])
        want[0] += row; want[1 + dom[b]] += row
    np.testing.assert_allclose(stats.reshape(1 + D, 11), want, rtol=1e-8, atol=1e-10)
    mean_lp, t_ee, t_ed = tr.step(lr, reg, reg_param_ua_scale=scale)
    np.testing.assert_allclose(mean_lp, want[0, 9] / B, rtol=1e-10)
    th0_ee, th0_ed = inputs['theta_en_en'].reshape(-1), inputs['theta_en_de'].reshape(-1)
    np.testing.assert_allclose(t_ee.cpu().numpy(), th0_ee + lr * (want[0, :3] - B * reg * th0_ee), rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(t_ed.cpu().numpy(), th0_ed + lr * (want[0, 3:9] - B * reg * th0_ed), rtol=1e-8, atol=1e-11)
    for d in range(D):
        np.testing.assert_allclose(tr.theta_dom_en_en[d].cpu().numpy(),
                                   th_dom_ee[d] + lr * (want[1 + d, :3] - sizes[d] * reg * scale * th_dom_ee[d]), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(tr.theta_dom_en_de[d].cpu().numpy(),
                                   th_dom_ed[d] + lr * (want[1 + d, 3:9] - sizes[d] * reg * scale * th_dom_ed[d]), rtol=1e-8, atol=1e-11)


def test_tidir_trainer_user_adapt(tmp_path):
    """TiDirTrainer(adapt='user'): domains from ti.user_id, per-domain rows in the params file (train_mp.py:80-102),
    one epoch from zero parameters against the oracle (all thetas zero -> every pot is 1: the per-domain sums are
    what is being checked), then a second epoch that must move the domain thetas apart."""
    import copy
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=30, X=64, Vde=64, sent_len=(4, 6), n_predicted=(1, 3), seed=9)
    tt = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                      paths['phi_ped'], sweeps=3, adapt='user', reg_param_ua_scale=0.5)
    assert len(tt.domains) > 1
    phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
    D = len(tt.domains)
    tot = np.zeros((1 + D, 11))
    feat = {'correct': 2, 'full_history': 3, 'hit_history': 4}
    th_ee, th_ed = np.zeros((1, 3)), np.zeros((1, 6))
    for key, b in sorted(tt.buckets.items()):
        unary = [f for f in sorted(b['spec']['factors'], key=lambda f: f['id']) if len(f['vars']) == 1]
        roots = [key[1][i % len(key[1])] for i in range(3)]
        for i, r in enumerate(b['rows']):
            s = copy.deepcopy(b['spec'])
            s['labels'] = [int(v) for v in b['var_labels'][i]]
            for u, f in enumerate(unary):
                s['factors'][f['id']]['observed_dim'] = int(b['unary_obs'][i, u])
            phi_i = phi_ed.copy()
            for name, k in feat.items():
                plane = np.zeros((64, 64))
                for ci, cj, cv in r['planes'][name]:
                    plane[ci, cj] += cv
                phi_i[:, :, k] = plane
            inp = _pots(dict(phi_en_en=phi_ee, phi_en_en_w1=phi_w1, phi_en_de=phi_i), th_ee, th_ed)
            g = O.Graph(s); msgs = O.init_messages(g)
            O.treelike_inference(g, inp, msgs, 3, roots, O.has_loops(g, roots[0]))
            ee, ed = O.unregularized_gradient(g, inp, msgs)
            row = np.concatenate([ee.reshape(-1), ed.reshape(-1), [O.log_posterior(g, msgs), 1.0]])
            tot[0] += row; tot[1 + tt.domains.index(str(r['user_id']))] += row
    save = os.path.join(str(tmp_path), 'params')
    hist = tt.train(epochs=1, reg_param=0.2, save_params=save)
    assert os.path.exists(save + '.user_adapt.iter0')            # train_mp.py:653-654: the adapt mode is part of the file name
    save = save + '.user_adapt'
    np.testing.assert_allclose(hist[0], tot[0, 9] / 30, rtol=1e-9)
    np.testing.assert_allclose(tt.theta_en_de.cpu().numpy(), 0.1 * tot[0, 3:9], rtol=1e-8, atol=1e-11)
    for d in range(D):
        np.testing.assert_allclose(tt.theta_dom_en_en[d].cpu().numpy(), 0.1 * tot[1 + d, :3], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(tt.theta_dom_en_de[d].cpu().numpy(), 0.1 * tot[1 + d, 3:9], rtol=1e-8, atol=1e-11)
    _, _, _, _, d2t = tidir.read_params(save)
    assert set(d2t) == {(ft, d) for ft in ('en_en', 'en_de') for d in tt.domains}
    np.testing.assert_allclose(d2t['en_de', tt.domains[0]].reshape(-1), 0.1 * tot[1, 3:9], atol=1e-6)
    before = tt.theta_dom_en_de.clone()
    tt.epoch(0.1, 0.2 / 30)
    assert not torch.equal(before, tt.theta_dom_en_de)
    assert float((tt.theta_dom_en_de[0] - tt.theta_dom_en_de[1]).abs().max()) > 0


def test_tidir_trainer_sweeps_all_sentence_shapes_in_one_launch(tmp_path):
    """TiDirTrainer(grouped_sweeps=True): the sweeps of every bucket (its own topology and roots) run as ONE launch
    sequence of the shared-table matrix-core kernels (mlbp_sweep_groups_f64 -> a group table), each bucket's gradient
    launch follows.  Same statistics as the per-bucket launch sequences, and training moves theta the same way."""
    from macaronicusermodeling_amd import _ffi, tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=40, X=64, Vde=64, sent_len=(4, 7), n_predicted=(2, 3), seed=21)
    mk = lambda grouped: TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                                      paths['phi_ped'], sweeps=3, grouped_sweeps=grouped)
    a, b = mk(True), mk(False)
    assert len(a.buckets) > 2
    for t in (a, b):                                   # away from theta = 0, where every pot is the all-ones table
        t.theta_en_en += torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=t.theta_en_en.device)
        t.theta_en_de += torch.tensor([0.2, 0.1, -0.3, 0.05, 0.0, 0.1], dtype=torch.float64, device=t.theta_en_de.device)
    sa = a.local_statistics().clone()
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3       # MLBP_KERNEL_SHARED_MFMA ran the groups
    assert _ffi.lib.mlbp_last_sweep_fused_gradient() == 1   # ... and each group's gradient as its epilogue
    assert sum(tr.batch.program(tr.roots[:tr.n_sweeps_run]).exact_count(tr.batch.B) for tr in a._full.trainers.values()) == 0
    sb = b.local_statistics().clone()
    np.testing.assert_allclose(sa.cpu().numpy(), sb.cpu().numpy(), rtol=1e-9, atol=1e-12)
    ha, hb = a.train(epochs=2, reg_param=0.2), b.train(epochs=2, reg_param=0.2)
    np.testing.assert_allclose(ha, hb, rtol=1e-9)
    np.testing.assert_allclose(a.theta_en_de.cpu().numpy(), b.theta_en_de.cpu().numpy(), rtol=1e-8, atol=1e-12)


def test_tidir_trainer_groups_with_larger_cliques_share_the_launch(tmp_path):
    """Sentences with up to four predicted words: the K4 buckets (six pairwise factors, three-source updates) share ONE prepare
    launch with the K2 / K3 buckets and run their own sweep launch beside theirs (one launch per form of the kernel, the
    product-fused one on a side stream; gradient epilogue included, the whole step replayed from one HIP graph).  Same
    statistics and the same training as per-bucket launches."""
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=36, X=64, Vde=64, sent_len=(5, 8), n_predicted=(2, 4), seed=33)
    mk = lambda grouped: TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                                      paths['phi_ped'], sweeps=3, grouped_sweeps=grouped)
    a, b = mk(True), mk(False)
    assert {tr.topo.P for tr in a._full.trainers.values()} >= {1, 3, 6}
    for t in (a, b):
        t.theta_en_en += torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=t.theta_en_en.device)
        t.theta_en_de += torch.tensor([0.2, 0.1, -0.3, 0.05, 0.0, 0.1], dtype=torch.float64, device=t.theta_en_de.device)
    from macaronicusermodeling_amd import _ffi
    sa = a.local_statistics().cpu().numpy()
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3 and _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    np.testing.assert_allclose(sa, b.local_statistics().cpu().numpy(), rtol=1e-9, atol=1e-12)
    ha, hb = a.train(epochs=3, reg_param=0.2), b.train(epochs=3, reg_param=0.2)       # (three epochs: replayed from a HIP graph)
    np.testing.assert_allclose(ha, hb, rtol=1e-9)
    np.testing.assert_allclose(a.theta_en_de.cpu().numpy(), b.theta_en_de.cpu().numpy(), rtol=1e-8, atol=1e-12)
    # theta far out: constant products past 1e280 -- the matrix-core kernels flag such graphs, the exact kernel redoes them, and
    # the K4 ones get their gradient from the per-graph kernel (grouped: ONE launch over every group's flagged graphs, sixteen
    # graphs of a group per block).  Finite, and the same statistics either way (the replayed graphs read theta from the device).
    for t in (a, b):         # (the bias plane of the en_en features: every en_en entry e^200, four given words' unary factors overflow a product)
        t.theta_en_en.copy_(torch.tensor([0.3, -0.2, 200.0], dtype=torch.float64, device=t.theta_en_en.device))
    sa2, sb2 = a.local_statistics().cpu().numpy(), b.local_statistics().cpu().numpy()
    redone = {tr.topo.P: 0 for tr in a._full.trainers.values()}
    for tr in a._full.trainers.values():
        redone[tr.topo.P] += tr.batch.program(tr.roots[:tr.n_sweeps_run]).exact_count(tr.batch.B)
    assert redone[6] > 0, redone
    assert np.isfinite(sa2).all()
    np.testing.assert_allclose(sa2, sb2, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('case', C.approx_cases(), ids=lambda c: c['name'])
def test_batched_approximate_inference_and_beliefs(case):
    """use_approx_inference / use_approx_beliefs batched (FactorGraphBatch(use_approx_*=True)): the top-100 selection
    runs on the device inside the sweep launch (one launch for all sweeps) and inside the gradient kernel -- no
    per-update round trip as in the object API.  Graph 0 against the fixture the reference itself produced with both
    switches on (messages, marginals, gradient: the selected index SETS must be the reference's for these to agree),
    the other graphs against the oracle in approximate mode."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = case['spec']
    gold = load_golden(case['name'])
    topo = GraphTopology.from_spec(spec)
    nb = 4
    inputs = case_inputs(case, nb)
    has_phi = 'phi_en_en' in inputs[0]
    if has_phi:
        for i in inputs[1:]:
            for k in ('phi_en_en', 'phi_en_en_w1', 'phi_en_de'):
                i[k] = inputs[0][k]
    pair, unary = batch_tables(spec, topo, inputs)
    fb = FactorGraphBatch(topo, spec['X'], nb, use_approx_inference=True, use_approx_beliefs=True)
    fb.set_pair_tables(pair)
    fb.set_unary_tables(unary)
    roots = case['roots'][:max(case['snaps'])]
    marg = torch.empty(nb, topo.n_vars, spec['X'], dtype=torch.float64, device=fb.device)
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 4 and prog.status() == 0        # the wide kernel, selection fused in
    got = fb.msgs.cpu().numpy()
    np.testing.assert_allclose(got[0], gold['msgs_s%d' % max(case['snaps'])], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(marg[0].cpu().numpy(), gold['marginals'], rtol=1e-10, atol=1e-300)
    g = O.Graph(spec)
    for b in range(1, nb):
        msgs = O.init_messages(g)
        for r in roots:
            O.sweep(g, inputs[b], msgs, r, True)
        want = np.stack([msgs[k] for k in C.msg_keys(spec)]).reshape(got[b].shape)
        np.testing.assert_allclose(got[b], want, rtol=1e-10, atol=1e-300)
    if has_phi:
        pair_phi, kinds, obs, labels = _meta(spec, topo)
        fb.set_features(inputs[0]['phi_en_en'], inputs[0]['phi_en_en_w1'], inputs[0]['phi_en_de'], pair_phi, kinds)
        fb.set_observations(np.tile(labels, (nb, 1)), np.tile(obs, (nb, 1)))
        g_ee, g_ed = fb.gradient()
        assert _ffi.lib.mlbp_gradient_status() == 0
        np.testing.assert_allclose(g_ee[0].cpu().numpy(), gold['grad_unreg_en_en'].reshape(-1), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(g_ed[0].cpu().numpy(), gold['grad_unreg_en_de'].reshape(-1), rtol=1e-8, atol=1e-11)
        for b in range(1, nb):
            msgs = O.init_messages(g)
            for r in roots:
                O.sweep(g, inputs[b], msgs, r, True)
            ee, ed = O.unregularized_gradient(g, inputs[b], msgs, True)
            np.testing.assert_allclose(g_ee[b].cpu().numpy(), ee.reshape(-1), rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(g_ed[b].cpu().numpy(), ed.reshape(-1), rtol=1e-8, atol=1e-11)
    # too few states for a top-100: the reference's argpartition raises; so does the batched switch
    small = FactorGraphBatch(GraphTopology.from_spec(C.ring_spec(4, 64)), 64, 2, use_approx_inference=True)
    small.set_pair_tables(np.ones((8, 64, 64))); small.set_unary_tables(np.ones((8, 64)))
    with pytest.raises(_ffi.MlbpError, match='out of bounds'):
        small.sweep([0], init=True)


def test_ti_dir_to_posterior_equals_the_reference_pipeline(tmp_path):
    """SURVEY 8 rows f1 + f3 end to end, pinned: tests/golden/tidir_reference.json holds, for 12 synthetic instances
    (X = 16, all three feature planes on, seeded theta), the marginals and log-posteriors the REFERENCE computes with its own
    TrainingInstance.from_dict -> create_factor_graph (train_mp.py:105-306: potentials from phi . theta with the per-instance
    planes written into phi_en_de, factor creation) -> initialize -> three sweeps -> get_posterior_probs
    (make_tidir_golden.py), and each instance's step `return_gradient()`.  TiDirTrainer must reach the same numbers from the
    same files, and one epoch must move theta by the sum of those steps."""
    import json
    import os
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_reference.json'), encoding='utf8'))
    d = str(tmp_path)
    paths = {k: os.path.join(d, k) for k in ('ti', 'vocab.en', 'vocab.de', 'phi.pmi', 'phi.pmi_w1', 'phi.ed', 'phi.ped')}
    open(paths['ti'], 'w', encoding='utf8').write('\n'.join(gold['instances']) + '\n')
    open(paths['vocab.en'], 'w', encoding='utf8').write('\n'.join(gold['vocab_en']) + '\n')
    open(paths['vocab.de'], 'w', encoding='utf8').write('\n'.join(gold['vocab_de']) + '\n')
    for k, name in (('phi.pmi', 'phi_pmi'), ('phi.pmi_w1', 'phi_pmi_w1'), ('phi.ed', 'phi_ed'), ('phi.ped', 'phi_ped')):
        np.savetxt(paths[k], np.array(gold[name]))
    assert gold['ee_names'] == tidir.EE_NAMES and gold['ed_names'] == tidir.ED_NAMES
    tt = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                      paths['phi.ped'], sweeps=3, use_correct_feat=True, history=True, session_history=True)
    tt.theta_en_en.copy_(torch.tensor(gold['theta_en_en'], dtype=torch.float64).reshape(-1))
    tt.theta_en_de.copy_(torch.tensor(gold['theta_en_de'], dtype=torch.float64).reshape(-1))
    by_sent = {json.loads(l)['current_sent'][0]['sent_id']: r for l, r in zip(gold['instances'], gold['reference'])}
    seen = 0
    for key, tr in tt.trainers.items():
        assert list(tr.roots) == by_sent[tt.buckets[key]['rows'][0]['sent_id']]['roots']
        lp, _, _, _ = tr.predict(top=min(16, tr.batch.X))
        marg = tr._marg.cpu().numpy()
        for i, row in enumerate(tt.buckets[key]['rows']):
            ref = by_sent[row['sent_id']]
            np.testing.assert_allclose(marg[i], np.array(ref['marginals']), rtol=1e-9, atol=1e-300)
            np.testing.assert_allclose(lp[i], ref['log_posterior'], rtol=1e-9)
            seen += 1
    assert seen == len(gold['reference']) == 12
    # one epoch = theta + the sum of the instances' steps at that theta (batch_sgd -> batch_sgd_accumulate,
    # train_mp.py:360-424; the gradient includes the three per-instance feature planes)
    o = gold['options']
    mean_lp = tt.epoch(o['learning_rate'], o['reg_param'] / len(gold['reference']))
    want_ee = np.array(gold['theta_en_en']).reshape(-1) + sum(np.array(r['step'][0]) for r in gold['reference'])
    want_ed = np.array(gold['theta_en_de']).reshape(-1) + sum(np.array(r['step'][1]) for r in gold['reference'])
    np.testing.assert_allclose(tt.theta_en_en.cpu().numpy(), want_ee, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tt.theta_en_de.cpu().numpy(), want_ed, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mean_lp, np.mean([r['log_posterior'] for r in gold['reference']]), rtol=1e-9)


@pytest.mark.parametrize('mode', ['user', 'experience'])
def test_ti_dir_adapt_epoch_equals_the_reference_batch_sgd(tmp_path, mode):
    """--user_adapt / --experience_adapt, pinned: tidir_reference.json['user_adapt' | 'experience_adapt'] holds what the
    reference's own batch_sgd (train_mp.py:360-398, run by make_tidir_golden.py) returns per instance with seeded per-domain
    thetas: potentials from the domain's theta INSTEAD of the global one, the global step, and the per-domain step
    regularised with reg_param_ua_scale (0.5 / 2.0).  One epoch of TiDirTrainer(adapt=mode) must add exactly the sums
    (batch_sgd_accumulate, train_mp.py:405-416)."""
    import json
    import os
    from macaronicusermodeling_amd.train import TiDirTrainer
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_reference.json'), encoding='utf8'))
    ua = gold['%s_adapt' % mode]
    names = ua['users'] if mode == 'user' else ua['domains']
    d = str(tmp_path)
    paths = {k: os.path.join(d, k) for k in ('ti', 'vocab.en', 'vocab.de', 'phi.pmi', 'phi.pmi_w1', 'phi.ed', 'phi.ped')}
    open(paths['ti'], 'w', encoding='utf8').write('\n'.join(gold['instances']) + '\n')
    open(paths['vocab.en'], 'w', encoding='utf8').write('\n'.join(gold['vocab_en']) + '\n')
    open(paths['vocab.de'], 'w', encoding='utf8').write('\n'.join(gold['vocab_de']) + '\n')
    for k, name in (('phi.pmi', 'phi_pmi'), ('phi.pmi_w1', 'phi_pmi_w1'), ('phi.ed', 'phi_ed'), ('phi.ped', 'phi_ped')):
        np.savetxt(paths[k], np.array(gold[name]))
    tt = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                      paths['phi.ped'], sweeps=3, adapt=mode, domains=names, reg_param_ua_scale=ua['reg_param_ua_scale'])
    tt.theta_en_en.copy_(torch.tensor(gold['theta_en_en'], dtype=torch.float64).reshape(-1))
    tt.theta_en_de.copy_(torch.tensor(gold['theta_en_de'], dtype=torch.float64).reshape(-1))
    for i, u in enumerate(names):
        tt.theta_dom_en_en[i].copy_(torch.tensor(ua['theta_dom'][u][0], dtype=torch.float64))
        tt.theta_dom_en_de[i].copy_(torch.tensor(ua['theta_dom'][u][1], dtype=torch.float64))
    o = gold['options']
    mean_lp = tt.epoch(o['learning_rate'], o['reg_param'] / len(gold['instances']))
    inst = ua['instances']
    np.testing.assert_allclose(mean_lp, np.mean([r['log_posterior'] for r in inst]), rtol=1e-9)
    np.testing.assert_allclose(tt.theta_en_en.cpu().numpy(), np.array(gold['theta_en_en']).reshape(-1) + sum(np.array(r['step'][0]) for r in inst),
                               rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tt.theta_en_de.cpu().numpy(), np.array(gold['theta_en_de']).reshape(-1) + sum(np.array(r['step'][1]) for r in inst),
                               rtol=1e-9, atol=1e-12)
    for i, u in enumerate(names):
        mine = [r for r in inst if r['user' if mode == 'user' else 'domain'] == u]
        assert mine
        np.testing.assert_allclose(tt.theta_dom_en_en[i].cpu().numpy(), np.array(ua['theta_dom'][u][0]) + sum(np.array(r['step_domain'][0]) for r in mine),
                                   rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(tt.theta_dom_en_de[i].cpu().numpy(), np.array(ua['theta_dom'][u][1]) + sum(np.array(r['step_domain'][1]) for r in mine),
                                   rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('shared', [False, True])
def test_gradient_fused_into_a_sweep_that_skips_unchanged_updates(shared):
    """MLBP_SWEEP_SKIP_UNCHANGED with the gradient in the same call: the fused gradient reads the final messages and the
    resident tables, so it must equal the full schedule's (to rounding: these kernels fuse adjacent updates), with
    per-graph tables (the scale-free kernel's epilogue) and with shared pots (separate gradient kernel after the MFMA sweep)."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(10, [1, 4, 7], 64, 48, seed=1)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 77)
    B, roots = 40, [4, 1, 7]
    labels, obs = _instances(spec, topo, B, 5)
    outs = []
    for skip in (False, True):
        tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                              inputs['theta_en_en'], inputs['theta_en_de'], roots=roots)
        tr.batch.skip_unchanged = skip
        tr.batch.pair_tables_shared = shared
        stats = tr.local_statistics().clone()
        assert tr.batch.program(tr.roots[:tr.n_sweeps_run]).skippable_updates() > 0
        assert _ffi.lib.mlbp_last_sweep_kernel() == (3 if shared else 7)
        outs.append(stats.cpu().numpy())
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-11, atol=1e-13)


def test_grouped_statistics_and_the_large_state_step_replay_from_a_hip_graph(tmp_path):
    """Every hot-path entry only enqueues (after one eager pass that makes the first-use allocations): the trainer's grouped
    sweeps (mlbp_sweep_groups_f64: group table owned by the first program, uploaded only when it changes) and the X = 128
    step (update-by-update MFMA contractions; the gradient's slot arrays come from the caller's host copy) are captured into
    HIP graphs and replayed -- same bits as the eager launches, also after theta has moved."""
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer, UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    paths = tidir.synthesize(str(tmp_path), n_instances=40, X=64, Vde=64, sent_len=(4, 7), n_predicted=(2, 3), seed=21)
    tr = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                      paths['phi_ped'], sweeps=3, grouped_sweeps=True)
    bump_ee = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=tr.theta_en_en.device)
    bump_ed = torch.tensor([0.2, 0.1, -0.3, 0.05, 0.0, 0.1], dtype=torch.float64, device=tr.theta_en_de.device)
    tr.theta_en_en += bump_ee; tr.theta_en_de += bump_ed
    eager1 = tr.local_statistics().clone()
    tr.capture()
    assert torch.equal(tr.local_statistics(), eager1)
    tr.theta_en_en += 0.5 * bump_ee; tr.theta_en_de -= 0.25 * bump_ed          # the graph reads theta from the same tensors
    replay2 = tr.local_statistics().clone()
    tr._graph = None
    assert torch.equal(tr.local_statistics(), replay2) and not torch.equal(replay2, eager1)
    # X = 128: shared pots, sweeps and pairwise gradient as batched contractions
    spec = C.user_spec(8, [1, 3, 6], 128, 40, seed=2)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 78)
    labels, obs = _instances(spec, topo, 6, 6)
    ut = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'], roots=[3, 1, 6])
    e1 = ut.local_statistics().clone()
    ut.capture()
    assert torch.equal(ut.local_statistics(), e1)
    ut.theta_en_en *= 0.5
    r2 = ut.local_statistics().clone()
    ut._graph = None
    assert torch.equal(ut.local_statistics(), r2) and not torch.equal(r2, e1)


def _large_state_trainer(B, seed, X=128):
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(8, [1, 3, 6], X, 40, seed=2)
    topo = GraphTopology.from_spec(spec)
    inputs = C.make_inputs(spec, 78)
    labels, obs = _instances(spec, topo, B, seed)
    return UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                            inputs['theta_en_en'], inputs['theta_en_de'], roots=[3, 1, 6])


def test_a_captured_large_state_step_survives_larger_calls_behind_it():
    """the X >= 128 shared-table path kept its fragment and gradient buffers in process-wide statics that a
    later, larger call freed and re-allocated -- a HIP graph captured before replayed against freed memory.  The buffers now
    belong to the program (and, for a standalone mlbp_gradient_f64, to the caller's workspace or a block that only grows by
    new allocations): a step captured at B = 64 replays bit-identically after an eager step at B = 512 and after standalone
    gradient calls of both sizes."""
    small = _large_state_trainer(64, 6)
    want = small.local_statistics().clone()
    small.capture()
    assert torch.equal(small.local_statistics(), want)
    big = _large_state_trainer(512, 7)
    big_stats = big.local_statistics().clone()              # larger B: more scratch everywhere
    big.batch.sweep(big.roots, init=True)
    big.batch.gradient()                                    # standalone, no workspace: the process-wide block grows
    small.batch.gradient()
    torch.cuda.synchronize()
    assert torch.equal(small.local_statistics(), want)      # replayed
    assert torch.equal(big.local_statistics(), big_stats)
    # a caller-owned workspace gives the same numbers as the fallback block
    from macaronicusermodeling_amd import _ffi
    import ctypes
    fb = big.batch
    fb.sweep(big.roots, init=True)
    ee1, ed1 = fb.gradient()
    ee2, ed2 = torch.empty_like(ee1), torch.empty_like(ed1)
    a = fb._gradient_args(ee2, ed2)
    need = _ffi.lib.mlbp_gradient_workspace_bytes(ctypes.byref(a))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=fb.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _ffi.check(_ffi.lib.mlbp_gradient_f64(ctypes.byref(a), None if False else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(ee1, ee2) and torch.equal(ed1, ed2)
    a.workspace_bytes = need - 8
    with pytest.raises(_ffi.MlbpError):
        _ffi.check(_ffi.lib.mlbp_gradient_f64(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def test_two_threads_run_large_state_steps_on_two_streams():
    """Different programs on different streams, X = 128 shared tables (sweeps as batched contractions + the pairwise gradient):
    every scratch buffer of the path belongs to a program, so two host threads with their own trainers and streams run
    concurrently and get what each gets alone (the header's concurrency rule; the buffers used to be process-wide)."""
    import threading
    alone = []
    for k in range(2):
        tr = _large_state_trainer(96 + 32 * k, 11 + k)
        alone.append(tr.local_statistics().clone())
    torch.cuda.synchronize()
    out, errs = [None, None], []

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                tr = _large_state_trainer(96 + 32 * k, 11 + k)
                for _ in range(6):
                    got = tr.local_statistics()
                st.synchronize()
                out[k] = got.clone()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert torch.equal(out[k], alone[k])


def _write_tidir(gold, d):
    import os
    paths = {k: os.path.join(d, k) for k in ('ti', 'vocab.en', 'vocab.de', 'phi.pmi', 'phi.pmi_w1', 'phi.ed', 'phi.ped')}
    open(paths['ti'], 'w', encoding='utf8').write('\n'.join(gold['instances']) + '\n')
    open(paths['vocab.en'], 'w', encoding='utf8').write('\n'.join(gold['vocab_en']) + '\n')
    open(paths['vocab.de'], 'w', encoding='utf8').write('\n'.join(gold['vocab_de']) + '\n')
    for k, name in (('phi.pmi', 'phi_pmi'), ('phi.pmi_w1', 'phi_pmi_w1'), ('phi.ed', 'phi_ed'), ('phi.ped', 'phi_ped')):
        np.savetxt(paths[k], np.array(gold[name]))
    return paths


def _batch_gold():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_batch_reference.json'), encoding='utf8'))


def _trainer(paths, gold, **kw):
    from macaronicusermodeling_amd.train import TiDirTrainer
    tt = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                      paths['phi.ped'], sweeps=3, **kw)
    if gold is not None:
        tt.theta_en_en.copy_(torch.tensor(gold['theta_en_en'], dtype=torch.float64).reshape(-1))
        tt.theta_en_de.copy_(torch.tensor(gold['theta_en_de'], dtype=torch.float64).reshape(-1))
    return tt


def test_minibatched_shuffled_epoch_equals_the_reference_sequence(tmp_path):
    """train_mp.py:631-649 + 405-424 shuffles the instances and updates theta as results come back.  tidir_batch_reference.json
    ['minibatch'] holds the REFERENCE's own batch_sgd run minibatch by minibatch (4 instances each, a fixed shuffled order,
    every instance of a minibatch at the theta the minibatch starts from, steps added as batch_sgd_accumulate adds them;
    make_batch_golden.py).  TiDirTrainer(minibatch=4) walking the same order must land on the same theta after every
    minibatch -- shapes are mixed inside a minibatch (grouped sweeps and per-bucket launches both), one update per minibatch.
    Both minibatch forms: 'masked' (round 4, the default: the whole shard resident, every minibatch a device-side selection of
    its instances' statistics -- eager, and replayed from ONE HIP graph) and 'rebuild' (bucket trainers built per minibatch)."""
    gold = _batch_gold()
    mb = gold['minibatch']
    paths = _write_tidir(gold, str(tmp_path))
    for grouped, mode, graph in ((True, 'rebuild', False), (False, 'rebuild', False), (True, 'masked', False), (False, 'masked', False),
                                 (True, 'masked', True)):
        tt = _trainer(paths, gold, minibatch=mb['size'], shuffle_seed=3, grouped_sweeps=grouped, minibatch_mode=mode)
        tt.epoch_order = lambda epoch: np.array(mb['order'])
        seen = []
        if mode == 'rebuild':
            update = tt._update

            def recording(lr, reg):
                out = update(lr, reg)
                seen.append((tt.theta_en_en.cpu().numpy().copy(), tt.theta_en_de.cpu().numpy().copy(), out))
                return out
            tt._update = recording
        else:
            update = tt._update_on_device

            def recording(lr, reg):
                update(lr, reg)
                n = tt.n_stat
                seen.append((tt.theta_en_en.cpu().numpy().copy(), tt.theta_en_de.cpu().numpy().copy(),
                             (float(tt.stats[n - 2].item()), float(tt.stats[n - 1].item()))))
            tt._update_on_device = recording
            if graph:
                tt.capture_masked()
        mean_lp = tt.epoch(mb['learning_rate'], gold['options']['reg_param'] / len(gold['instances']))
        assert len(seen) == len(mb['steps']) == 3
        for (ee, ed, (lp, n)), step in zip(seen, mb['steps']):
            np.testing.assert_allclose(ee, step['theta_en_en'], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(ed, step['theta_en_de'], rtol=1e-9, atol=1e-12)
            assert n == len(step['instances'])
            np.testing.assert_allclose(lp, sum(step['log_posteriors']), rtol=1e-9)
        np.testing.assert_allclose(mean_lp, np.mean([l for s in mb['steps'] for l in s['log_posteriors']]), rtol=1e-9)
    # a seeded shuffle: every epoch another order, the same on every rank, each a permutation of all instances
    o0, o1 = tt.__class__.epoch_order(tt, 0), tt.__class__.epoch_order(tt, 1)
    assert sorted(o0) == sorted(o1) == list(range(12)) and list(o0) != list(o1)
    assert list(tt.__class__.epoch_order(_trainer(paths, gold, minibatch=4, shuffle_seed=3), 0)) == list(o0)


def test_batched_prediction_files_equal_the_reference_text(tmp_path):
    """--save_predictions (train_mp.py:740-760): per instance the '*SENT_ID:' block (FactorGraph.to_string, LBP.py:109-123) and
    the .dist lines (to_dist, LBP.py:125-143).  TiDirTrainer.predict(save_predictions=...) writes both files from the batched
    top-50 indices and log-marginals; they must equal, character for character, what the reference's batch_predictions
    returned for the same 12 instances (tidir_batch_reference.json['predictions']), and the precision counts and mean
    log-posterior the sums of its per-instance values."""
    gold = _batch_gold()
    paths = _write_tidir(gold, str(tmp_path))
    tt = _trainer(paths, gold)
    out = str(tmp_path / 'pred')
    mean_lp, counts = tt.predict(save_predictions=out)
    want = gold['predictions']
    assert open(out, encoding='utf8').read() == ''.join(p['block'] + '\n' for p in want)
    assert open(out + '.dist', encoding='utf8').read() == ''.join(p['dist'] + '\n' for p in want)
    assert counts == tuple(int(sum(p['precision'][k] for p in want)) for k in range(4))
    np.testing.assert_allclose(mean_lp, np.mean([p['log_posterior'] for p in want]), rtol=1e-9)


@pytest.mark.parametrize('adapt', [None, 'user'])
def test_resume_from_a_params_file(tmp_path, adapt):
    """--load_params (train_mp.py:528-542): a run that starts from the file another run saved continues from its thetas --
    global and, with --user_adapt, every user's -- and the adapt mode's extension is tried before the bare name."""
    gold = _batch_gold()
    paths = _write_tidir(gold, str(tmp_path))
    a = _trainer(paths, gold, adapt=adapt, minibatch=5, shuffle_seed=1)
    a.train(epochs=2, reg_param=0.2, save_params=str(tmp_path / 'params'))
    ext = '.user_adapt' if adapt else ''
    import os
    assert os.path.exists(str(tmp_path / 'params') + ext) and os.path.exists(str(tmp_path / 'params') + ext + '.iter1')
    b = _trainer(paths, None, adapt=adapt, minibatch=5, shuffle_seed=1, load_params=str(tmp_path / 'params'))
    b._epochs_done = 2                                   # (walks the third epoch's order, as the first run does next)
    # the file keeps six decimals (save_params, train_mp.py:80-102: '%0.6f'): that is what a resumed run starts from
    r6 = lambda t: np.round(t.cpu().numpy(), 6)
    np.testing.assert_allclose(b.theta_en_en.cpu().numpy(), r6(a.theta_en_en), rtol=0, atol=1e-12)
    np.testing.assert_allclose(b.theta_en_de.cpu().numpy(), r6(a.theta_en_de), rtol=0, atol=1e-12)
    b2 = _trainer(paths, None, adapt=adapt)
    b2.load_params(str(tmp_path / 'params'))
    np.testing.assert_array_equal(b2.theta_en_de.cpu().numpy(), b.theta_en_de.cpu().numpy())
    if adapt:
        assert b.domains == a.domains and len(a.domains) > 1
        np.testing.assert_allclose(b.theta_dom_en_en.cpu().numpy(), r6(a.theta_dom_en_en), rtol=0, atol=1e-12)
        np.testing.assert_allclose(b.theta_dom_en_de.cpu().numpy(), r6(a.theta_dom_en_de), rtol=0, atol=1e-12)
        assert float(a.theta_dom_en_de.abs().sum()) > 0
    # the resumed run moves on from there as the first one would (to the file's precision)
    la, lb = a.epoch(0.05, 0.2 / 12), b.epoch(0.05, 0.2 / 12)
    np.testing.assert_allclose(lb, la, rtol=1e-4)


def _rank_worker(rank, world, port, paths, gold_path, q):
    """One fresh rank process of the two-rank trainer test: gloo group, both ranks on cuda:0."""
    import json
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch as th
    from macaronicusermodeling_amd import dist as mdist
    from macaronicusermodeling_amd.train import TiDirTrainer
    r, w, _ = mdist.init_from_env(backend='gloo')
    th.cuda.set_device(0)
    gold = json.load(open(gold_path, encoding='utf8'))
    ua = gold['user_adapt']
    tt = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                      paths['phi.ped'], sweeps=3, adapt='user', domains=ua['users'], reg_param_ua_scale=ua['reg_param_ua_scale'],
                      rank=r, world=w)
    tt.theta_en_en.copy_(th.tensor(gold['theta_en_en'], dtype=th.float64).reshape(-1))
    tt.theta_en_de.copy_(th.tensor(gold['theta_en_de'], dtype=th.float64).reshape(-1))
    for i, u in enumerate(ua['users']):
        tt.theta_dom_en_en[i].copy_(th.tensor(ua['theta_dom'][u][0], dtype=th.float64))
        tt.theta_dom_en_de[i].copy_(th.tensor(ua['theta_dom'][u][1], dtype=th.float64))
    shapes = sorted(str(k) for k in tt.trainers)
    o = gold['options']
    mean_lp = tt.epoch(o['learning_rate'], o['reg_param'] / len(gold['instances']))
    pred_lp, counts = tt.predict()
    q.put((rank, shapes, mean_lp, tt.theta_en_en.cpu().numpy(), tt.theta_en_de.cpu().numpy(), tt.theta_dom_en_en.cpu().numpy(),
           tt.theta_dom_en_de.cpu().numpy(), pred_lp, counts, sum(t.batch.B for t in tt.trainers.values())))
    th.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_trainer_epoch_equals_one_process_and_the_reference(tmp_path):
    """The multi-rank path of the trainer itself (train_mp.py:634-649 + 405-424 as contiguous shards + ONE all-reduce of the
    fused [global | per-domain] statistics buffer): two FRESH rank processes (gloo, both on cuda:0 -- the one-GPU box's
    rehearsal of two GPUs) run TiDirTrainer.epoch with --user_adapt on the 12-instance golden TI_DIR.  The shards hold
    different sentence shapes.  Every rank must end with the thetas -- global and every user's -- of the single-process run
    and of the reference's summed batch_sgd steps (tidir_reference.json), and predict() must return the all-instance totals
    on every rank."""
    import json
    import os
    import socket
    import torch.multiprocessing as mp
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_reference.json')
    gold = json.load(open(gold_path, encoding='utf8'))
    paths = _write_tidir(gold, str(tmp_path))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, paths, gold_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1] and res[0][9] + res[1][9] == 12          # different shape buckets, all instances covered
    ua, inst = gold['user_adapt'], gold['user_adapt']['instances']
    want_ee = np.array(gold['theta_en_en']).reshape(-1) + sum(np.array(r['step'][0]) for r in inst)
    want_ed = np.array(gold['theta_en_de']).reshape(-1) + sum(np.array(r['step'][1]) for r in inst)
    for r in res:
        np.testing.assert_allclose(r[3], want_ee, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r[4], want_ed, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r[2], np.mean([x['log_posterior'] for x in inst]), rtol=1e-9)
        for i, u in enumerate(ua['users']):
            mine = [x for x in inst if x['user'] == u]
            np.testing.assert_allclose(r[5][i], np.array(ua['theta_dom'][u][0]) + sum(np.array(x['step_domain'][0]) for x in mine), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(r[6][i], np.array(ua['theta_dom'][u][1]) + sum(np.array(x['step_domain'][1]) for x in mine), rtol=1e-9, atol=1e-12)
    for k in (3, 4, 5, 6):
        np.testing.assert_array_equal(res[0][k], res[1][k])                 # every rank holds the same parameters
    assert res[0][7] == res[1][7] and res[0][8] == res[1][8] and res[0][8][3] > 0      # prediction totals reduced
    # ... and the single-process trainer gets there too (same numbers to rounding: other summation order)
    from macaronicusermodeling_amd.train import TiDirTrainer
    one = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                       paths['phi.ped'], sweeps=3, adapt='user', domains=ua['users'], reg_param_ua_scale=ua['reg_param_ua_scale'])
    one.theta_en_en.copy_(torch.tensor(gold['theta_en_en'], dtype=torch.float64).reshape(-1))
    one.theta_en_de.copy_(torch.tensor(gold['theta_en_de'], dtype=torch.float64).reshape(-1))
    for i, u in enumerate(ua['users']):
        one.theta_dom_en_en[i].copy_(torch.tensor(ua['theta_dom'][u][0], dtype=torch.float64))
        one.theta_dom_en_de[i].copy_(torch.tensor(ua['theta_dom'][u][1], dtype=torch.float64))
    o = gold['options']
    one.epoch(o['learning_rate'], o['reg_param'] / len(gold['instances']))
    np.testing.assert_allclose(res[0][3], one.theta_en_en.cpu().numpy(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(res[0][6], one.theta_dom_en_de.cpu().numpy(), rtol=1e-11, atol=1e-13)
    lp1, c1 = one.predict()
    assert c1 == res[0][8]
    np.testing.assert_allclose(res[0][7], lp1, rtol=1e-11)


def _rank_worker_predict(rank, world, port, paths, gold_path, out, q):
    """One fresh rank process: the prediction pass with --save_predictions, thetas of tidir_batch_reference.json."""
    import json
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch as th
    from macaronicusermodeling_amd import dist as mdist
    from macaronicusermodeling_amd.train import TiDirTrainer
    r, w, _ = mdist.init_from_env(backend='gloo')
    th.cuda.set_device(0)
    gold = json.load(open(gold_path, encoding='utf8'))
    tt = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                      paths['phi.ped'], sweeps=3, rank=r, world=w)
    tt.theta_en_en.copy_(th.tensor(gold['theta_en_en'], dtype=th.float64).reshape(-1))
    tt.theta_en_de.copy_(th.tensor(gold['theta_en_de'], dtype=th.float64).reshape(-1))
    q.put((rank,) + tuple(tt.predict(save_predictions=out)))
    th.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_write_one_prediction_file_in_instance_order(tmp_path):
    """train_mp.py:740-760 writes ONE '<save_predictions><ext>' and one '.dist', in instance order, and eval.py:32-60 /
    get_acc.py:27-46 read exactly those.  Two rank processes each write their shard; rank 0 joins the shards (contiguous, so rank
    order is instance order) and removes them.  The joined files equal the reference's text for the 12 golden instances
    character for character, and both ranks return the all-instance mean and counts."""
    import os
    import socket
    import torch.multiprocessing as mp
    gold = _batch_gold()
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_batch_reference.json')
    paths = _write_tidir(gold, str(tmp_path))
    out = str(tmp_path / 'pred')
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker_predict, args=(r, 2, port, paths, gold_path, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = gold['predictions']
    assert open(out, encoding='utf8').read() == ''.join(p['block'] + '\n' for p in want)
    assert open(out + '.dist', encoding='utf8').read() == ''.join(p['dist'] + '\n' for p in want)
    assert not [f for f in os.listdir(str(tmp_path)) if '.rank' in f]           # the shards are gone
    for r in res:
        assert r[2] == tuple(int(sum(p['precision'][k] for p in want)) for k in range(4))
        np.testing.assert_allclose(r[1], np.mean([p['log_posterior'] for p in want]), rtol=1e-9)


def test_tune_set_is_evaluated_after_every_epoch(tmp_path):
    """train_mp.py:657-684: after every epoch `batch_predictions` runs over the --tune instances under the thetas the epoch
    left, and the mean log-posterior and precision at 0 / 25 / 50 are reported.  TiDirTrainer.train(tune=path) keeps them in
    tune_history; here the tune file is the 12-instance golden TI_DIR itself, so that -- started from the golden thetas with a
    zero learning-rate schedule replaced by one real epoch -- (i) the evaluation BEFORE any update equals the reference's
    batch_predictions totals (tidir_batch_reference.json) and (ii) every epoch's entry equals what a fresh trainer loaded with
    that epoch's saved params file predicts."""
    from macaronicusermodeling_amd.train import TiDirTrainer
    gold = _batch_gold()
    paths = _write_tidir(gold, str(tmp_path))
    want = gold['predictions']
    tt = _trainer(paths, gold)
    tuner = tt.tune_evaluator(paths['ti'])
    assert tuner.theta_en_en is tt.theta_en_en and tuner.theta_en_de is tt.theta_en_de          # reads the trainer's tensors
    lp0, c0 = tuner.predict()
    assert c0 == tuple(int(sum(p['precision'][k] for p in want)) for k in range(4))
    np.testing.assert_allclose(lp0, np.mean([p['log_posterior'] for p in want]), rtol=1e-9)
    save = str(tmp_path / 'params')
    hist = tt.train(epochs=2, reg_param=0.2, save_params=save, tune=tuner, capture=False)
    assert len(hist) == 2 and len(tt.tune_history) == 2 and tt.tune_history[0] != tt.tune_history[1]
    for epoch in range(2):
        fresh = TiDirTrainer(paths['ti'], paths['vocab.en'], paths['vocab.de'], paths['phi.pmi'], paths['phi.pmi_w1'], paths['phi.ed'],
                             paths['phi.ped'], sweeps=3, load_params='%s.iter%d' % (save, epoch))
        lp, c = fresh.predict()
        assert c == tt.tune_history[epoch][1]
        np.testing.assert_allclose(tt.tune_history[epoch][0], lp, rtol=1e-4)           # (the params file keeps six decimals)
    # a tune FILE is accepted as well
    t2 = _trainer(paths, gold)
    t2.train(epochs=1, reg_param=0.2, tune=paths['ti'], capture=False)
    np.testing.assert_allclose(t2.tune_history[0][0], tt.tune_history[0][0], rtol=1e-9)


def test_single_predicted_word_takes_its_gradient_in_the_sweep_launch():
    """A sentence with ONE predicted word has no pairwise factor (train_mp.py:257-299: the complete graph over one variable): the
    exact X = 64 kernel computes its marginal and -- fused -- its gradient, unary terms only, in one launch (it used to be two: the
    sweep launch and the per-graph gradient kernel).  Against the standalone gradient kernel and the oracle."""
    import copy
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.train import UserGraphTrainer
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.user_spec(9, [4], 64, 48, seed=3)
    topo = GraphTopology.from_spec(spec)
    assert topo.P == 0
    inputs = C.reference_planes(C.make_inputs(spec, 41))
    B = 45
    labels, obs = _instances(spec, topo, B, 9)
    tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                          inputs['theta_en_en'], inputs['theta_en_de'])
    fb = tr.batch
    tr.build_potentials()
    roots = tr.roots[:tr.n_sweeps_run]
    g_ee, g_ed = torch.full_like(tr._g_ee, float('nan')), torch.full_like(tr._g_ed, float('nan'))
    fb.sweep(roots, init=True, marginals=tr._marg, gradient=(g_ee, g_ed), keep_messages=False)
    assert _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    marg = tr._marg.clone()
    fb.sweep(roots, init=True, marginals=tr._marg)
    assert _ffi.lib.mlbp_last_sweep_fused_gradient() == 0
    assert torch.equal(marg, tr._marg)
    h_ee, h_ed = fb.gradient()
    np.testing.assert_allclose(g_ee.cpu().numpy(), h_ee.cpu().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(g_ed.cpu().numpy(), h_ed.cpu().numpy(), rtol=1e-12, atol=1e-13)
    for b in range(0, B, 11):
        s = copy.deepcopy(spec)
        g0 = O.Graph(s)
        s['labels'] = [int(labels[b, g0.var_order.index(v)]) for v in s['var_ids']]
        unary_ids = [f['id'] for f in g0.factors if len(f['vars']) == 1]
        for f in s['factors']:
            f['observed_dim'] = int(obs[b, unary_ids.index(f['id'])])
        g = O.Graph(s)
        msgs = O.init_messages(g)
        O.treelike_inference(g, inputs, msgs, len(roots), list(roots), O.has_loops(g, roots[0]))
        ee, ed = O.unregularized_gradient(g, inputs, msgs)
        np.testing.assert_allclose(g_ee[b].cpu().numpy(), ee.reshape(-1), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g_ed[b].cpu().numpy(), ed.reshape(-1), rtol=1e-9, atol=1e-12)


def test_tidir_trainer_groups_take_single_word_sentences_along(tmp_path):
    """Sentences with ONE predicted word (no pairwise factor) ride on the grouped call of the other shapes: the library flags all
    their graphs, and the exact kernel's pass over every group's flagged graphs and the flagged graphs' gradient -- launches the
    call makes anyway -- do their work.  Same statistics and the same training as per-bucket launches (replayed from HIP graphs)."""
    from macaronicusermodeling_amd import _ffi, tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    paths = tidir.synthesize(str(tmp_path), n_instances=48, X=64, Vde=64, sent_len=(4, 7), n_predicted=(1, 3), seed=52)
    mk = lambda grouped: TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'],
                                      paths['phi_ped'], sweeps=3, grouped_sweeps=grouped)
    a, b = mk(True), mk(False)
    ps = {tr.topo.P for tr in a._full.trainers.values()}
    assert 0 in ps and ps & {1, 3}
    for t in (a, b):
        t.theta_en_en += torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=t.theta_en_en.device)
        t.theta_en_de += torch.tensor([0.2, 0.1, -0.3, 0.05, 0.0, 0.1], dtype=torch.float64, device=t.theta_en_de.device)
    sa = a.local_statistics().cpu().numpy()
    assert _ffi.lib.mlbp_last_sweep_kernel() == 3 and _ffi.lib.mlbp_last_sweep_fused_gradient() == 1
    np.testing.assert_allclose(sa, b.local_statistics().cpu().numpy(), rtol=1e-9, atol=1e-12)
    ha, hb = a.train(epochs=3, reg_param=0.2), b.train(epochs=3, reg_param=0.2)
    np.testing.assert_allclose(ha, hb, rtol=1e-9)
    np.testing.assert_allclose(a.theta_en_en.cpu().numpy(), b.theta_en_en.cpu().numpy(), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(a.theta_en_de.cpu().numpy(), b.theta_en_de.cpu().numpy(), rtol=1e-8, atol=1e-12)
    # predictions (marginals of every shape) agree as well
    pa, pb = a.predict(), b.predict()
    np.testing.assert_allclose(pa[0], pb[0], rtol=1e-9)
