"""Seeded case definitions shared by the golden-vector generator, the oracle tests and the GPU parity tests.

A *spec* is plain data describing one factor graph in the vocabulary of the reference
(`LBP.py`): variables with ids / supervised label indices and factors in CREATION order (the order
in which `FactorNode.add_varset_with_potentials` and `FactorGraph.add_factor` are called, which
fixes `VariableNode.facset` order, LBP.py:449-452, and `FactorGraph.variables` insertion order,
LBP.py:149-153).  `build_graph(L, spec, inputs)` drives ANY module `L` that exposes the `LBP.py`
object API -- the translated reference (generator only), or the product module -- through exactly
the same calls, so fixtures, oracle and product are compared on identical graphs.

Nothing here is derived from reference source text; the two construction styles mirror the call
patterns of run.py:41-56 ("explicit") and train_mp.py:257-305 ("trainmp").
"""
import numpy as np

F_EE = 3  # len(theta_en_en_names), train_mp.py:520
F_ED = 6  # len(theta_en_de_names), train_mp.py:521-523


# ------------------------------------------------------------------------------------------------
# topologies
# ------------------------------------------------------------------------------------------------
def chain_spec(n, X, name=None):
    """n unary factors (ids 0..n-1) then n-1 pairwise (ids n..2n-2) over (i, i+1); run.py:43-51."""
    factors = []
    for i in range(n):
        factors.append(dict(id=i, vars=[i], dims=[0], table=i))
    for i in range(n - 1):
        factors.append(dict(id=n + i, vars=[i, i + 1], dims=[0, 1], table=n + i))
    return dict(name=name or 'chain%d_x%d' % (n, X), style='explicit', X=X,
                var_ids=list(range(n)), labels=[i % X for i in range(n)], factors=factors)


def ring_spec(n, X, name=None):
    """chain + closing factor over (0, n-1) with var 0 on dim 0; run.py:53-56."""
    s = chain_spec(n, X, name or 'ring%d_x%d' % (n, X))
    s['factors'].append(dict(id=2 * n - 1, vars=[0, n - 1], dims=[0, 1], table=2 * n - 1))
    return s


def star_spec(n_leaves, X, name=None):
    """hub 0 with pairwise factors to leaves 1..n; leaves carry a unary factor.  The hub sits on
    dim 1 of odd factors to exercise both orientations and a varset order that differs from the
    dim order."""
    factors = []
    fid = 0
    for i in range(1, n_leaves + 1):
        factors.append(dict(id=fid, vars=[i], dims=[0], table=fid)); fid += 1
    for i in range(1, n_leaves + 1):
        if i % 2:
            factors.append(dict(id=fid, vars=[i, 0], dims=[0, 1], table=fid))
        else:
            factors.append(dict(id=fid, vars=[i, 0], dims=[1, 0], table=fid))
        fid += 1
    return dict(name=name or 'star%d_x%d' % (n_leaves, X), style='explicit', X=X,
                var_ids=list(range(n_leaves + 1)), labels=[(3 * i + 1) % X for i in range(n_leaves + 1)],
                factors=factors)


def shuffled_ids_spec(X, name=None):
    """A small loopy graph whose factor ids are NOT in creation order and whose variable ids are
    sparse, so `initialize`'s sort-by-id (LBP.py:195-196) and dict-keyed variables matter."""
    var_ids = [7, 2, 11, 5]
    fac = [
        dict(id=9, vars=[7, 2], dims=[0, 1], table=0),
        dict(id=3, vars=[2], dims=[0], table=1),
        dict(id=6, vars=[11, 2], dims=[0, 1], table=2),
        dict(id=1, vars=[7, 11], dims=[1, 0], table=3),
        dict(id=4, vars=[5, 11], dims=[0, 1], table=4),
        dict(id=8, vars=[5], dims=[0], table=5),
        dict(id=0, vars=[7], dims=[0], table=6),
    ]
    return dict(name=name or 'shuffled_x%d' % X, style='explicit', X=X, var_ids=var_ids,
                labels=[1 % X, 0, 3 % X, 2 % X], factors=fac)


def user_spec(sent_len, predicted, X, Vde, name=None, seed=0):
    """train_mp.py:257-299 shape: per predicted var one unary `en_de` factor (gap 0); per predicted
    pair one pairwise `en_en` (gap = |id diff|, dim0 = lower id); per (predicted, given) pair one
    unary `en_en` factor observed at the given word's index.  Given variables own no factors and
    never enter the graph.  Variable id = sentence index."""
    rs = np.random.RandomState(1000 + seed)
    predicted = sorted(predicted)
    labels_all = [int(v) for v in rs.randint(0, X, size=sent_len)]       # guess / given word index in en
    de_obs = [int(v) for v in rs.randint(0, Vde, size=sent_len)]          # l2 word index in de
    factors = []
    for i in range(sent_len):
        if i in predicted:
            factors.append(dict(id=len(factors), vars=[i], dims=[0], factor_type='en_de', gap=0,
                                observed_dim=de_obs[i], obs_size=Vde, position=i))
    for a in range(sent_len):
        for b in range(a + 1, sent_len):
            pa, pb = a in predicted, b in predicted
            if pa and pb:
                factors.append(dict(id=len(factors), vars=[a, b], dims=[0, 1], factor_type='en_en',
                                    gap=abs(a - b), observed_dim=None, obs_size=None, position=None))
            elif not pa and not pb:
                continue
            else:
                g, p = (a, b) if pb else (b, a)
                factors.append(dict(id=len(factors), vars=[p], dims=[0], factor_type='en_en',
                                    gap=abs(g - p), observed_dim=labels_all[g], obs_size=X, position=g))
    return dict(name=name or 'user_p%d_g%d_x%d' % (len(predicted), sent_len - len(predicted), X),
                style='trainmp', X=X, Vde=Vde, var_ids=list(predicted),
                labels=[labels_all[i] for i in predicted], factors=factors)


# ------------------------------------------------------------------------------------------------
# seeded inputs
# ------------------------------------------------------------------------------------------------
def make_inputs(spec, seed, table_kind='uniform'):
    """Float inputs for a spec.  explicit: one table per distinct `table` key, U(0,1)+0.01
     or exp(N(0,1)).  trainmp: phi tensors, theta rows and the pots
    exp(phi . theta^T) exactly as train_mp.py:220-255 forms them."""
    rs = np.random.RandomState(seed)
    X = spec['X']
    if spec['style'] == 'explicit':
        ntab = 1 + max(f['table'] for f in spec['factors'])
        shapes = {}
        for f in spec['factors']:
            shapes[f['table']] = (X, X) if len(f['vars']) == 2 else (X, 1)
        tables = []
        for t in range(ntab):
            if table_kind == 'uniform':
                tables.append(rs.rand(*shapes[t]) + 0.01)
            else:
                tables.append(np.exp(rs.randn(*shapes[t])))
        return dict(tables=tables)
    Vde = spec['Vde']
    phi_en_en = rs.rand(X, X, F_EE)
    phi_en_en_w1 = rs.rand(X, X, F_EE)
    phi_en_de = rs.rand(X, Vde, F_ED)
    theta_en_en = rs.randn(1, F_EE) * 0.5
    theta_en_de = rs.randn(1, F_ED) * 0.5
    pot_en_en = np.exp(phi_en_en.dot(theta_en_en.T).reshape(X, X))
    pot_en_en_w1 = np.exp(phi_en_en_w1.dot(theta_en_en.T).reshape(X, X))
    pot_en_de = np.exp(phi_en_de.dot(theta_en_de.T).reshape(X, Vde))
    return dict(phi_en_en=phi_en_en, phi_en_en_w1=phi_en_en_w1, phi_en_de=phi_en_de,
                theta_en_en=theta_en_en, theta_en_de=theta_en_de,
                pot_en_en=pot_en_en, pot_en_en_w1=pot_en_en_w1, pot_en_de=pot_en_de)


def reference_planes(inputs):
    """The en_en feature tensors as train_mp.py:600-606 stacks them -- phi_en_en = [pmi, 0, 1], phi_en_en_w1 = [pmi, pmi_w1, 1] --
    on make_inputs' random values, pots redone."""
    inputs = dict(inputs)
    ee, w1 = inputs['phi_en_en'].copy(), inputs['phi_en_en_w1'].copy()
    ee[:, :, 1] = 0.0; ee[:, :, 2] = 1.0; w1[:, :, 0] = ee[:, :, 0]; w1[:, :, 2] = 1.0
    X = ee.shape[0]
    inputs.update(phi_en_en=ee, phi_en_en_w1=w1, pot_en_en=np.exp(ee.dot(inputs['theta_en_en'].T).reshape(X, X)),
                  pot_en_en_w1=np.exp(w1.dot(inputs['theta_en_en'].T).reshape(X, X)))
    return inputs


def domain_of(X):
    return ['w%d' % i for i in range(X)]


# ------------------------------------------------------------------------------------------------
# drive an LBP-API module
# ------------------------------------------------------------------------------------------------
def build_graph(L, spec, inputs):
    """Construct the graph through the `LBP.py` object API of module L."""
    X = spec['X']
    dom = domain_of(X)
    if spec['style'] == 'trainmp':
        fg = L.FactorGraph(['ee%d' % i for i in range(F_EE)], ['ed%d' % i for i in range(F_ED)],
                           inputs['theta_en_en'].copy(), inputs['theta_en_de'].copy(),
                           inputs['phi_en_en_w1'], inputs['phi_en_en'], inputs['phi_en_de'])
        fg.pot_en_en = inputs['pot_en_en']
        fg.pot_en_en_w1 = inputs['pot_en_en_w1']
        fg.pot_en_de = inputs['pot_en_de']
    else:
        z = np.zeros((1, 1))
        fg = L.FactorGraph([], [], z, z, None, None, None)
    vs = {}
    for vid, lab in zip(spec['var_ids'], spec['labels']):
        vs[vid] = L.VariableNode(vid, L.VAR_TYPE_PREDICTED, 'en', dom, dom[lab])
    facs = []
    for f in spec['factors']:
        v2d = dict(zip(f['vars'], f['dims']))
        if spec['style'] == 'trainmp':
            fn = L.FactorNode(f['id'], factor_type=f['factor_type'],
                              observed_domain_size=f['obs_size'])
            pt = L.PotentialTable(v_id2dim=v2d, table=None, observed_dim=f['observed_dim'])
            fn.add_varset_with_potentials(varset=[vs[v] for v in f['vars']], ptable=pt)
            fn.position = f['position']
            fn.gap = f['gap']
        else:
            fn = L.FactorNode(f['id'])
            pt = L.PotentialTable(v_id2dim=v2d, table=inputs['tables'][f['table']])
            fn.add_varset_with_potentials(varset=[vs[v] for v in f['vars']], ptable=pt)
        facs.append(fn)
    for fn in facs:
        fg.add_factor(fn)
    if spec['style'] == 'trainmp':
        for fn in fg.factors:
            fn.potential_table.slice_potentials()
    return fg


def msg_keys(spec):
    """Canonical message order used in fixtures: factors sorted by id (LBP.py:195-196); unary ->
    (F,v); pairwise -> for each var in varset order (v,F) then (F,v) (LBP.py:211-216)."""
    keys = []
    for f in sorted(spec['factors'], key=lambda d: d['id']):
        fname = 'F_%d' % f['id']
        if len(f['vars']) == 1:
            keys.append((fname, 'X_%d' % f['vars'][0]))
        else:
            for v in f['vars']:
                keys.append(('X_%d' % v, fname))
                keys.append((fname, 'X_%d' % v))
    return keys


def node_code(name):
    """'X_5' -> (0, 5); 'F_3' -> (1, 3)."""
    return (0 if name[0] == 'X' else 1, int(name[2:]))


# ------------------------------------------------------------------------------------------------
# the case list
# ------------------------------------------------------------------------------------------------
def inference_cases():
    """(case name, spec, input seed, table kind, root sequence, snapshot sweeps, force_loopy)."""
    cases = []
    ch = chain_spec(8, 64)
    cases.append(dict(name='chain8_x64_forced', spec=ch, seed=1235, kind='uniform',
                      roots=[0] * 10, snaps=[1, 2, 3, 10], force_loopy=True))
    cases.append(dict(name='chain8_x64_natural', spec=ch, seed=1235, kind='uniform',
                      roots=[3], snaps=[1], force_loopy=False, request=10))
    rg = ring_spec(8, 64)
    cases.append(dict(name='ring8_x64', spec=rg, seed=1235, kind='uniform',
                      roots=[0] * 10, snaps=[1, 2, 3, 10], force_loopy=False))
    cases.append(dict(name='ring8_x64_roots', spec=rg, seed=77, kind='lognormal',
                      roots=[5, 2, 7, 0, 3], snaps=[1, 3, 5], force_loopy=False))
    cases.append(dict(name='star5_x4', spec=star_spec(5, 4), seed=5, kind='uniform',
                      roots=[0, 2], snaps=[1, 2], force_loopy=True))
    cases.append(dict(name='shuffled_x4', spec=shuffled_ids_spec(4), seed=6, kind='uniform',
                      roots=[11, 5, 7], snaps=[1, 2, 3], force_loopy=False))
    cases.append(dict(name='shuffled_x64', spec=shuffled_ids_spec(64), seed=8, kind='lognormal',
                      roots=[2, 7, 5], snaps=[3], force_loopy=False))
    u3 = user_spec(10, [1, 4, 7], 64, 64, seed=1)
    cases.append(dict(name='user_k3_x64', spec=u3, seed=1236, kind=None,
                      roots=[1, 4, 7], snaps=[1, 2, 3], force_loopy=False))
    u3w = user_spec(10, [2, 3, 6], 64, 48, name='user_k3w1_x64', seed=2)
    cases.append(dict(name='user_k3w1_x64', spec=u3w, seed=1237, kind=None,
                      roots=[6, 2, 3], snaps=[3], force_loopy=False))
    u2 = user_spec(6, [0, 3], 64, 64, seed=3)
    cases.append(dict(name='user_k2_x64', spec=u2, seed=21, kind=None,
                      roots=[3], snaps=[1], force_loopy=False, request=3))
    u1 = user_spec(5, [2], 64, 32, seed=9)
    cases.append(dict(name='user_k1_x64', spec=u1, seed=23, kind=None,
                      roots=[2], snaps=[1], force_loopy=False, request=3))
    u4 = user_spec(10, [0, 2, 5, 6], 128, 128, seed=4)
    cases.append(dict(name='user_k4_x128', spec=u4, seed=22, kind=None,
                      roots=[0, 6, 2], snaps=[1, 3], force_loopy=False))
    u5 = user_spec(8, [0, 1, 3, 5, 7], 16, 12, seed=5)
    cases.append(dict(name='user_k5_x16', spec=u5, seed=24, kind=None,
                      roots=[7, 0, 3], snaps=[3], force_loopy=False))
    cases.append(dict(name='ring8_x512', spec=ring_spec(8, 512), seed=1239, kind='uniform',
                      roots=[0] * 10, snaps=[10], force_loopy=False, light=True))
    return cases


def approx_cases():
    """use_approx_inference / use_approx_beliefs variants (top-K=100 needs X >= 100)."""
    u3 = user_spec(10, [1, 4, 7], 128, 128, name='user_k3_x128', seed=11)
    return [dict(name='approx_user_k3_x128', spec=u3, seed=31, kind=None, roots=[1, 4, 7], snaps=[3],
                 force_loopy=False),
            dict(name='approx_ring6_x128', spec=ring_spec(6, 128), seed=32, kind='lognormal',
                 roots=[0, 3, 5], snaps=[3], force_loopy=False)]


def schedule_topologies():
    out = [chain_spec(8, 4), ring_spec(8, 4), ring_spec(3, 4), star_spec(5, 4), shuffled_ids_spec(4)]
    for p, sl in [(1, 4), (2, 5), (3, 10), (4, 10), (5, 9)]:
        pred = list(range(0, 2 * p, 2))[:p]
        out.append(user_spec(sl, pred, 4, 4, name='sched_user_p%d_l%d' % (p, sl), seed=p))
    # two components: has_loops / schedules only see the root's component (LBP.py:174-190)
    s = chain_spec(3, 4, name='two_components')
    base = len(s['factors'])
    s['var_ids'] += [10, 11, 12]
    s['labels'] += [0, 1, 2]
    for i, (a, b) in enumerate([(10, 11), (11, 12), (10, 12)]):
        s['factors'].append(dict(id=100 + i, vars=[a, b], dims=[0, 1], table=base + i))
    out.append(s)
    return out
