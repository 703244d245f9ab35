"""GPU parity of the shared-table sweep kernel (float64 MFMA, 16 graphs per workgroup; mlbp_shared.hip)
against the CPU oracle: the reference's own table layout -- one pot_en_en / pot_en_en_w1 array behind every
pairwise factor (LBP.py:456-467), per-graph unary columns.  Tolerance 1e-10 relative (north star: 1e-5)."""
import numpy as np
import pytest

import cases as C
from helpers import oracle_msgs
from oracle import lbp_oracle as O

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

RTOL = 1e-10
KERNEL_SHARED_MFMA = 3


def _shared_batch(spec, B, seed=7, mutate=None):
    """B graphs of one topology: pairwise tables = the two en_en pots of graph 0 (shared by all), unary
    tables = each graph's own pots (seeded).  Returns (batch, topo, per-graph inputs)."""
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    topo = GraphTopology.from_spec(spec)
    X = spec['X']
    inputs = [C.make_inputs(spec, seed + 1000 * b) for b in range(B)]
    for b in range(1, B):
        inputs[b]['pot_en_en'] = inputs[0]['pot_en_en']
        inputs[b]['pot_en_en_w1'] = inputs[0]['pot_en_en_w1']
    if mutate:
        mutate(inputs)
    g = O.Graph(spec)
    by_id = {f['id']: f for f in spec['factors']}
    pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
    pair = np.stack([inputs[0]['pot_en_en'], inputs[0]['pot_en_en_w1']])
    unary = np.stack([O.factor_table(g, inputs[b], g.by_id[topo.factor_ids[j]]).reshape(X)
                      for b in range(B) for j in topo.unary_factors])
    fb = FactorGraphBatch(topo, X, B)
    fb.set_pair_tables(pair, np.tile(np.array(pair_phi), (B, 1)))
    fb.set_unary_tables(unary)
    assert fb.pair_tables_shared
    return fb, topo, inputs


SPECS = {
    'user_k3_gaps_3_6': lambda: C.user_spec(10, [1, 4, 7], 64, 64, seed=1),        # one distinct table
    'user_k3_gaps_1_2_3': lambda: C.user_spec(10, [1, 2, 4], 64, 64, seed=2),      # both en_en pots
    'user_k2': lambda: C.user_spec(6, [0, 1], 64, 64, seed=3),
    'user_k4': lambda: C.user_spec(9, [0, 2, 3, 7], 64, 64, seed=4),
    'user_k4_both_pots': lambda: C.user_spec(9, [0, 1, 3, 4], 64, 64, seed=5),   # two adjacent pairs: both en_en pots, every orientation
    'user_k5': lambda: C.user_spec(9, [0, 2, 3, 5, 8], 64, 64, seed=6),          # ten pairwise factors: the general form, most tiles in memory
    'user_k6': lambda: C.user_spec(8, [0, 1, 3, 4, 6, 7], 64, 64, seed=7),       # fifteen
}


@pytest.mark.parametrize('name', list(SPECS))
@pytest.mark.parametrize('B', [1, 37])
def test_shared_table_kernel_matches_oracle(name, B):
    from macaronicusermodeling_amd import _ffi
    spec = SPECS[name]()
    fb, topo, inputs = _shared_batch(spec, B)
    roots = [v for v in topo.var_ids][:3]
    roots = (roots * 3)[:3]
    marg = torch.full((B, topo.n_vars, 64), float('nan'), dtype=torch.float64, device=fb.device)
    fb.msgs.fill_(float('nan'))
    prog = fb.sweep(roots, init=True, marginals=marg)
    # user_k4: the three-source product-fused form (12 message tiles + 5 stored variable->factor messages in LDS, one workgroup per CU)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA, _ffi.lib.mlbp_last_error()
    assert prog.status() == 0 and prog.exact_count(B) == 0
    got, gm = fb.msgs.cpu().numpy(), marg.cpu().numpy()
    for b in range(B):
        g, msgs, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)
        for k, v in enumerate(topo.var_ids):
            np.testing.assert_allclose(gm[b, k], O.marginal(g, msgs, v).reshape(-1), rtol=RTOL, atol=1e-300)


def test_shared_kernel_agrees_with_exact_kernel_and_skips_writeback_on_request():
    from macaronicusermodeling_amd import _ffi
    spec = SPECS['user_k3_gaps_1_2_3']()
    B = 200
    fb, topo, _ = _shared_batch(spec, B)
    roots = [1, 2, 4]
    m1 = torch.empty(B, topo.n_vars, 64, dtype=torch.float64, device=fb.device)
    fb.sweep(roots, init=True, marginals=m1)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA
    msgs1 = fb.msgs.clone()
    try:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))
        m3 = torch.empty_like(m1)
        fb.sweep(roots, init=True, marginals=m3)
        assert _ffi.lib.mlbp_last_sweep_kernel() == 2
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    np.testing.assert_allclose(msgs1.cpu().numpy(), fb.msgs.cpu().numpy(), rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(m1.cpu().numpy(), m3.cpu().numpy(), rtol=1e-11, atol=1e-300)
    fb.msgs.fill_(-1.0)
    m2 = torch.empty_like(m1)
    fb.sweep(roots, init=True, marginals=m2, keep_messages=False)
    assert torch.equal(m2, m1)
    assert bool((fb.msgs == -1.0).all())          # nothing was written back


def test_degenerate_graphs_in_a_shared_batch_are_redone_exactly():
    """Zero-sum unary columns (uniform rule, LBP.py:655-657) and inf entries make the graph leave the MFMA
    kernel; the exact kernel recomputes exactly those graphs."""
    spec = SPECS['user_k3_gaps_3_6']()

    def mutate(inputs):
        inputs[5]['pot_en_de'] = inputs[5]['pot_en_de'].copy(); inputs[5]['pot_en_de'][:, :] = 0.0
        inputs[21]['pot_en_de'] = inputs[21]['pot_en_de'].copy(); inputs[21]['pot_en_de'][3, :] = np.inf
    B = 40
    fb, topo, inputs = _shared_batch(spec, B, mutate=mutate)
    roots = [1, 4, 7]
    marg = torch.empty(B, topo.n_vars, 64, dtype=torch.float64, device=fb.device)
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert prog.exact_count(B) == 2
    got = fb.msgs.cpu().numpy()
    with np.errstate(all='ignore'):
        for b in range(B):
            _, _, want = oracle_msgs(spec, inputs[b], roots)
            np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)


@pytest.mark.parametrize('name', ['user_k3_gaps_3_6', 'user_k3_gaps_1_2_3', 'user_k2', 'user_k4', 'user_k4_both_pots'])
def test_product_fused_form_with_zeros_in_the_constant_products(name):
    """The product-fused form of the kernel (K2, K3: the producer of a message stores c (.) message, the marginal c (.) m_a (.) m_b
    comes out as P_a (.) (P_b / c); K4: it stores sqrt(c) (.) message, a contraction's input is the product of two tiles and the
    marginal Q_a (.) Q_b (.) Q_c / sqrt(c)) on unary tables with exact zeros -- states where c is 0: the marginal is 0 there, the
    written-back MESSAGES are not (they come from the raw results the members stash) -- against the oracle, with and without the
    message write-back; and the host says the form applies."""
    from macaronicusermodeling_amd import _ffi
    spec = SPECS[name]()

    def mutate(inputs):
        rs = np.random.RandomState(7)
        for b, inp in enumerate(inputs):
            if b % 2 == 0:
                pot = inp['pot_en_de'].copy()
                pot[rs.rand(*pot.shape) < 0.3] = 0.0           # zeros scattered over every unary column (no column all zero)
                pot[0, :] = np.maximum(pot[0, :], 0.5)
                inp['pot_en_de'] = pot
    B = 37
    fb, topo, inputs = _shared_batch(spec, B, mutate=mutate)
    roots = ([v for v in topo.var_ids] * 3)[:3]
    assert topo.plan(roots)['shared_product_fused3' if name.startswith('user_k4') else 'shared_product_fused'] == 1
    marg = torch.full((B, topo.n_vars, 64), float('nan'), dtype=torch.float64, device=fb.device)
    fb.msgs.fill_(float('nan'))
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA, _ffi.lib.mlbp_last_error()
    assert prog.status() == 0 and prog.exact_count(B) == 0
    got, gm = fb.msgs.cpu().numpy(), marg.cpu().numpy()
    assert (gm == 0.0).any()                                    # the zeros are there
    for b in range(B):
        g, msgs, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)
        for k, v in enumerate(topo.var_ids):
            np.testing.assert_allclose(gm[b, k], O.marginal(g, msgs, v).reshape(-1), rtol=RTOL, atol=1e-300)
    m2 = torch.full_like(marg, float('nan'))
    fb.sweep(roots, init=True, marginals=m2, keep_messages=False)       # read-out only: nothing is stashed, the same marginals
    assert torch.equal(m2, marg)


def test_a_wrong_shared_claim_is_caught_on_the_device():
    """pair_tab rows that differ inside a group of 16 graphs (the caller's flag was wrong) send the group to
    the exact kernel; results are those of the tables each graph really points at."""
    spec = SPECS['user_k3_gaps_1_2_3']()
    B = 48
    fb, topo, inputs = _shared_batch(spec, B)
    tab = fb.pair_tab.cpu().numpy().copy()
    tab[20] = 1 - tab[20]                      # graph 20 swaps the two pots
    fb.pair_tab = torch.from_numpy(tab).to(fb.device)
    assert fb.pair_tables_shared               # stale claim, on purpose
    roots = [1, 2, 4]
    prog = fb.sweep(roots, init=True)
    assert prog.exact_count(B) == 16
    got = fb.msgs.cpu().numpy()
    for b in (0, 15, 32, 47):                  # graphs of the untouched groups: MFMA kernel, oracle values
        _, _, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)
    from macaronicusermodeling_amd import _ffi
    try:                                       # every graph, graph 20 included: what the exact kernel computes
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))
        fb.sweep(roots, init=True)
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    np.testing.assert_allclose(got, fb.msgs.cpu().numpy(), rtol=1e-11, atol=1e-300)
    assert not np.allclose(got[20], got[19])


def test_shared_table_gradient_matches_per_graph_kernels_and_oracle():
    """Train step in the reference's table layout: sweeps on the MFMA kernel, pairwise part of the gradient on
    the MFMA kernel, against (i) the per-graph gradient kernel on the same messages and (ii) the oracle."""
    from macaronicusermodeling_amd import _ffi
    spec = SPECS['user_k3_gaps_1_2_3']()
    B = 37
    fb, topo, inputs = _shared_batch(spec, B)
    by_id = {f['id']: f for f in spec['factors']}
    pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
    kinds = [2 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else (0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1)
             for j in topo.unary_factors]
    fb.set_features(inputs[0]['phi_en_en'], inputs[0]['phi_en_en_w1'], inputs[0]['phi_en_de'], pair_phi, kinds)
    labels = np.tile(np.array([dict(zip(spec['var_ids'], spec['labels']))[v] for v in topo.var_ids]), (B, 1))
    obs = np.tile(np.array([by_id[topo.factor_ids[j]]['observed_dim'] for j in topo.unary_factors]), (B, 1))
    fb.set_observations(labels, obs)
    roots = [1, 2, 4]
    gee = torch.full((B, 3), float('nan'), dtype=torch.float64, device=fb.device)
    ged = torch.full((B, 6), float('nan'), dtype=torch.float64, device=fb.device)
    fb.sweep(roots, init=True, gradient=(gee, ged))
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA and _ffi.lib.mlbp_gradient_status() == 0
    fb.use_shared_gradient = False                      # same messages, per-graph gradient kernel
    ee2, ed2 = fb.gradient()
    np.testing.assert_allclose(gee.cpu().numpy(), ee2.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ged.cpu().numpy(), ed2.cpu().numpy(), rtol=1e-9, atol=1e-12)
    for b in (0, 17, 36):
        for k in ('phi_en_en', 'phi_en_en_w1', 'phi_en_de'):
            inputs[b][k] = inputs[0][k]
        g, msgs, _ = oracle_msgs(spec, inputs[b], roots)
        want_ee, want_ed = O.unregularized_gradient(g, inputs[b], msgs)
        np.testing.assert_allclose(gee[b].cpu().numpy(), np.asarray(want_ee).reshape(-1), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(ged[b].cpu().numpy(), np.asarray(want_ed).reshape(-1), rtol=1e-8, atol=1e-10)


def test_full_size_properties_of_the_shared_kernel():
    """BASELINE batch size (8192 graphs): size-independent properties -- every message and marginal sums to 1,
    graphs with identical inputs get identical bits wherever they sit in the batch (different workgroups, different
    columns of the MFMA tile), and a sample of graphs equals the per-graph exact kernel to rounding."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = SPECS['user_k3_gaps_1_2_3']()
    topo = GraphTopology.from_spec(spec)
    B, X = 8192, 64
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    by_id = {f['id']: f for f in spec['factors']}
    which = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
    pair = torch.rand(2, X, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
    n_distinct = 517                                   # unary table sets repeat with a period coprime to 16
    unary = torch.rand(n_distinct * topo.U, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
    utab = (np.arange(B)[:, None] % n_distinct) * topo.U + np.arange(topo.U)[None, :]
    fb = FactorGraphBatch(topo, X, B)
    fb.set_pair_tables(pair, np.tile(np.array(which), (B, 1)))
    fb.set_unary_tables(unary, utab)
    roots = [1, 2, 4]
    marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA and prog.exact_count(B) == 0
    assert float((fb.msgs.sum(-1) - 1).abs().max()) < 1e-13 and float((marg.sum(-1) - 1).abs().max()) < 1e-13
    assert torch.equal(fb.msgs[:n_distinct], fb.msgs[n_distinct:2 * n_distinct])
    assert torch.equal(marg[:400], marg[15 * n_distinct:15 * n_distinct + 400])
    m1, g1 = fb.msgs[:64].clone(), marg[:64].clone()
    try:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))
        fb.sweep(roots, init=True, marginals=marg)
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    np.testing.assert_allclose(m1.cpu().numpy(), fb.msgs[:64].cpu().numpy(), rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(g1.cpu().numpy(), marg[:64].cpu().numpy(), rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize('X', [96, 128, 200, 333, 384, 512])
def test_large_state_shared_tables_run_as_batched_gemms(X):
    """X = 65 .. 512 with shared pairwise tables (any vocabulary size, X = len(en_domain) in train_mp.py:591-594: sizes that are
    no multiple of 128 run zero-padded, odd ones with 8-byte accesses): the sweeps run update by update over the whole batch, every
    factor->variable update one launch of the hand-written float64 MFMA contraction (mlbp_gemm.hip; the variable
    product fused into its prologue, Message.renormalize into its epilogue).  Against the oracle per graph and against the per-graph wide kernel on the
    same inputs (same updates, only the summation order inside the contraction differs)."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.ring_spec(5, X)
    topo = GraphTopology.from_spec(spec)
    B = 37                                      # three workgroups of 16 graphs, the last one ragged
    inputs = [C.make_inputs(spec, 31 + 1000 * b) for b in range(B)]
    g = O.Graph(spec)
    pair = np.stack([O.factor_table(g, inputs[0], g.by_id[topo.factor_ids[j]]).reshape(X, X) for j in topo.pair_factors])
    for b in range(1, B):                       # every graph: graph 0's pairwise tables, its own unary columns
        tabs = list(inputs[b]['tables'])
        for f in spec['factors']:
            if len(f['vars']) == 2:
                tabs[f['table']] = inputs[0]['tables'][f['table']]
        inputs[b] = dict(tables=tabs)
    unary = np.stack([O.factor_table(g, inputs[b], g.by_id[topo.factor_ids[j]]).reshape(X) for b in range(B) for j in topo.unary_factors])
    fb = FactorGraphBatch(topo, X, B)
    fb.set_pair_tables(pair, np.tile(np.arange(topo.P), (B, 1)))
    fb.set_unary_tables(unary)
    roots = [0, 3, 0]
    marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=fb.device)
    fb.msgs.fill_(float('nan'))
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 6, _ffi.lib.mlbp_last_error()      # MLBP_KERNEL_SHARED_GEMM
    assert prog.status() == 0
    got, gm = fb.msgs.clone(), marg.clone()
    try:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))                             # per-graph kernels on the same inputs
        fb.sweep(roots, init=True, marginals=marg)
        assert _ffi.lib.mlbp_last_sweep_kernel() == 4             # the wide kernel (X = 384: its padded instance)
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    np.testing.assert_allclose(got.cpu().numpy(), fb.msgs.cpu().numpy(), rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(gm.cpu().numpy(), marg.cpu().numpy(), rtol=1e-11, atol=1e-300)
    for b in (0, 17, B - 1):
        _, _, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b].cpu().numpy(), want, rtol=RTOL, atol=1e-300)
    # a false sharing statement is reported, not silently computed
    tab = fb.pair_tab.cpu().numpy().copy(); tab[4, 1] = tab[4, 0]
    fb.pair_tab = torch.from_numpy(tab).to(fb.device)
    prog = fb.sweep(roots, init=True)
    assert prog.status() == 2


@pytest.mark.parametrize('X', [640, 777, 1000, 1100, 1536, 2048])
def test_vocabulary_sized_shared_tables_run_as_batched_gemms(X):
    """X = len(en_domain) in the reference (train_mp.py:591-594) and its pairwise tables are always shared (LBP.py:456-467,
    695-706): sizes past 512 stay on the matrix cores.  Up to 1024 states the one-image contraction kernel (its LDS image of 16
    graphs is at most 131 KiB); beyond, the chunked kernel, which walks the table in blocks of at most 1024 x 1024 states
    (1100 -> 2 x 2 blocks of 640, 1536 -> 768, 2048 -> 1024).  Against the oracle per graph and against the per-graph
    kernels on the same inputs (the wide kernel up to 1024 states, the generic one beyond)."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.ring_spec(3, X)
    topo = GraphTopology.from_spec(spec)
    B = 19                                      # two workgroups of 16 graphs, the second one ragged
    rs = np.random.RandomState(4000 + X)
    shared = C.make_inputs(spec, 61)            # graph 0: the pairwise tables every graph reads
    inputs = []
    for b in range(B):                          # ... and its own unary columns
        tabs = list(shared['tables'])
        for f in spec['factors']:
            if len(f['vars']) == 1 and b > 0:
                tabs[f['table']] = rs.rand(X, 1) + 0.01
        inputs.append(dict(tables=tabs))
    g = O.Graph(spec)
    pair = np.stack([O.factor_table(g, inputs[0], g.by_id[topo.factor_ids[j]]).reshape(X, X) for j in topo.pair_factors])
    unary = np.stack([O.factor_table(g, inputs[b], g.by_id[topo.factor_ids[j]]).reshape(X) for b in range(B) for j in topo.unary_factors])
    fb = FactorGraphBatch(topo, X, B)
    fb.set_pair_tables(pair, np.tile(np.arange(topo.P), (B, 1)))
    fb.set_unary_tables(unary)
    roots = [0, 2, 1]
    marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=fb.device)
    fb.msgs.fill_(float('nan'))
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 6, _ffi.lib.mlbp_last_error()      # MLBP_KERNEL_SHARED_GEMM
    assert prog.status() == 0
    got, gm = fb.msgs.clone(), marg.clone()
    assert float((got.sum(-1) - 1).abs().max()) < 1e-12
    try:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(3))                             # per-graph kernels on the same inputs
        fb.sweep(roots, init=True, marginals=marg)
        assert _ffi.lib.mlbp_last_sweep_kernel() == (4 if X <= 1024 else 5)
    finally:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(1))
    np.testing.assert_allclose(got.cpu().numpy(), fb.msgs.cpu().numpy(), rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(gm.cpu().numpy(), marg.cpu().numpy(), rtol=1e-11, atol=1e-300)
    for b in (0, 15, B - 1):
        _, _, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b].cpu().numpy(), want, rtol=RTOL, atol=1e-300)


@pytest.mark.parametrize('which', ['ring8', 'chain8', 'chain2', 'chain3', 'ring3'])
def test_shared_kernel_with_most_tiles_spilled(which):
    """Rings / chains of 8 variables need 22-24 message tiles; 8-9 stay in LDS, the rest live in the global spill area
    (two distinct tables alternate along the factors).  Against the oracle per graph.  The short ones (chain of 2 or 3, ring of
    3) fit LDS whole and take the product-fused form: a chain's end variables have one pairwise factor, so the message into them
    is read by no update -- the kernel keeps the message itself for the read-out -- and their own message is the constant
    product alone."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    X, B = 64, 21
    n = int(which[-1])
    spec = C.ring_spec(n, X) if which.startswith('ring') else C.chain_spec(n, X)
    topo = GraphTopology.from_spec(spec)
    inputs = [C.make_inputs(spec, 51 + 1000 * b) for b in range(B)]
    g = O.Graph(spec)
    two = [O.factor_table(g, inputs[0], g.by_id[topo.factor_ids[j]]).reshape(X, X) for j in topo.pair_factors[:2]]
    pick = [p % 2 for p in range(topo.P)]
    for b in range(B):                          # every graph: the two shared tables alternating, its own unary columns
        tabs = list(inputs[b]['tables'])
        for p, j in enumerate(topo.pair_factors):
            tabs[g.by_id[topo.factor_ids[j]]['table']] = two[pick[p]]
        inputs[b] = dict(tables=tabs)
    unary = np.stack([O.factor_table(g, inputs[b], g.by_id[topo.factor_ids[j]]).reshape(X) for b in range(B) for j in topo.unary_factors])
    fb = FactorGraphBatch(topo, X, B)
    fb.set_pair_tables(np.stack(two), np.tile(np.array(pick), (B, 1)))
    fb.set_unary_tables(unary)
    roots = [0, 5 % n, 2 % n]
    assert topo.plan(roots)['shared_product_fused'] == 1       # (every variable here has at most two pairwise factors)
    marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=fb.device)
    fb.msgs.fill_(float('nan'))
    prog = fb.sweep(roots, init=True, marginals=marg)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA, _ffi.lib.mlbp_last_error()
    assert prog.status() == 0 and prog.exact_count(B) == 0
    got, gm = fb.msgs.cpu().numpy(), marg.cpu().numpy()
    for b in range(B):
        gg, msgs, want = oracle_msgs(spec, inputs[b], roots)
        np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)
        for k, v in enumerate(topo.var_ids):
            np.testing.assert_allclose(gm[b, k], O.marginal(gg, msgs, v).reshape(-1), rtol=RTOL, atol=1e-300)


@pytest.mark.parametrize('X', [256, 512])
def test_large_state_shared_float32_tables_on_the_f32_matrix_cores(X):
    """BASELINE config 5's "batched f32 MFMA message contraction": shared float32 tables at X = 256 / 512 run on
    v_mfma_f32_16x16x4_f32 (float32 table and message fragments, float32 sums over 64 states, float64 across them).
    Tolerance study (the numbers DESIGN.md 4.2b quotes): against the float64 oracle on the UNROUNDED tables the
    messages stay inside the north star's 1e-5 relative; against the float64 contraction fed the float32-rounded
    tables the difference is what the float32 message fragments and partial sums add."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology
    spec = C.ring_spec(6, X)
    topo = GraphTopology.from_spec(spec)
    B = 20
    inputs = [C.make_inputs(spec, 77 + 1000 * b, 'uniform') for b in range(B)]
    g = O.Graph(spec)
    pair = np.stack([O.factor_table(g, inputs[0], g.by_id[topo.factor_ids[j]]).reshape(X, X) for j in topo.pair_factors])
    for b in range(1, B):
        tabs = list(inputs[b]['tables'])
        for f in spec['factors']:
            if len(f['vars']) == 2:
                tabs[f['table']] = inputs[0]['tables'][f['table']]
        inputs[b] = dict(tables=tabs)
    unary = np.stack([O.factor_table(g, inputs[b], g.by_id[topo.factor_ids[j]]).reshape(X) for b in range(B) for j in topo.unary_factors])
    roots = [0, 2, 4, 0, 2, 4, 0, 2, 4, 0]            # ten sweeps, as config 5 runs
    idx = np.tile(np.arange(topo.P), (B, 1))
    fb32 = FactorGraphBatch(topo, X, B)
    fb32.set_pair_tables(pair, idx, dtype=torch.float32)
    fb32.set_unary_tables(unary)
    m32 = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=fb32.device)
    prog = fb32.sweep(roots, init=True, marginals=m32)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 6, _ffi.lib.mlbp_last_error()
    assert prog.status() == 0
    fb64 = FactorGraphBatch(topo, X, B)
    fb64.set_pair_tables(pair.astype(np.float32).astype(np.float64), idx)
    fb64.set_unary_tables(unary)
    m64 = torch.empty_like(m32)
    fb64.sweep(roots, init=True, marginals=m64)
    assert _ffi.lib.mlbp_last_sweep_kernel() == 6
    got32, got64 = fb32.msgs.cpu().numpy(), fb64.msgs.cpu().numpy()
    rel_arith = float(np.abs(got32 / got64 - 1).max())
    rel_marg = float(np.abs(m32.cpu().numpy() / m64.cpu().numpy() - 1).max())
    worst = 0.0
    for b in (0, 7, B - 1):
        _, _, want = oracle_msgs(spec, inputs[b], roots)
        worst = max(worst, float(np.abs(got32[b] / want.reshape(got32[b].shape) - 1).max()))
    print('X=%d f32 MFMA path: vs f64 contraction on rounded tables %.2e (messages) %.2e (marginals); vs unrounded oracle %.2e'
          % (X, rel_arith, rel_marg, worst))
    assert rel_arith < 2e-6 and rel_marg < 2e-6 and worst < 1e-5


def test_groups_of_mixed_shapes_in_one_shared_launch_sequence():
    """mlbp_sweep_groups_f64 over five topologies (K2, two K3, K4 -- three-source updates --, K5 -- the general form), each with its
    own roots, batch size (partial last groups of 16) and tables: ONE prepare launch and one sweep launch PER FORM of the
    shared-table kernel (product-fused: K2, K3; its three-source variant: K4; general: K5) behind a group table.  Same bits as the five
    single launch sequences, the oracle's values, a degenerate graph of one group redone by the exact kernel."""
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import sweep_groups
    names, sizes = ['user_k2', 'user_k3_gaps_1_2_3', 'user_k4', 'user_k3_gaps_3_6', 'user_k5'], [5, 37, 18, 16, 21]      # (K5: the general form, a third launch)

    def mutate(inputs):
        inputs[3]['pot_en_de'] = inputs[3]['pot_en_de'].copy(); inputs[3]['pot_en_de'][:, :] = 0.0
    built = [_shared_batch(SPECS[n](), B, seed=11 + k, mutate=mutate if k == 1 else None) for k, (n, B) in enumerate(zip(names, sizes))]
    roots = [(list(topo.var_ids) * 3)[k:k + 3] for k, (_, topo, _) in enumerate(built)]
    single_msgs, single_marg = [], []
    for (fb, topo, _), r in zip(built, roots):
        m = torch.empty(fb.B, topo.n_vars, 64, dtype=torch.float64, device=fb.device)
        fb.sweep(r, init=True, marginals=m)
        assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA
        single_msgs.append(fb.msgs.clone()); single_marg.append(m)
        fb.msgs.fill_(float('nan'))
    margs = [torch.full_like(m, float('nan')) for m in single_marg]
    progs = sweep_groups([fb for fb, _, _ in built], roots, init=True, marginals=margs)
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA, _ffi.lib.mlbp_last_error()
    assert [p.exact_count(B) for p, B in zip(progs, sizes)] == [0, 1, 0, 0, 0]
    for k, (fb, topo, inputs) in enumerate(built):
        # (every group runs the form of the kernel its single launch took: the same bits)
        assert torch.equal(fb.msgs, single_msgs[k]) and torch.equal(margs[k], single_marg[k])
        got = fb.msgs.cpu().numpy()
        with np.errstate(all='ignore'):
            for b in range(0, fb.B, 3):
                _, _, want = oracle_msgs(SPECS[names[k]](), inputs[b], roots[k])
                np.testing.assert_allclose(got[b], want, rtol=RTOL, atol=1e-300)
    # without the K4 group: one sweep launch, the same bits
    few = [0, 1, 3]
    for k in few:
        built[k][0].msgs.fill_(float('nan')); margs[k].fill_(float('nan'))
    sweep_groups([built[k][0] for k in few], [roots[k] for k in few], init=True, marginals=[margs[k] for k in few])
    assert _ffi.lib.mlbp_last_sweep_kernel() == KERNEL_SHARED_MFMA, _ffi.lib.mlbp_last_error()
    for k in few:
        assert torch.equal(built[k][0].msgs, single_msgs[k]) and torch.equal(margs[k], single_marg[k])
    # a second call with other batch contents re-uses the uploaded group table; one group that does not qualify
    # (no shared-table claim) sends the whole call to the per-group path
    built[0][0].pair_tables_shared = False
    sweep_groups([fb for fb, _, _ in built], roots, init=True, marginals=margs)
    for k, (fb, _, _) in enumerate(built):
        np.testing.assert_allclose(fb.msgs.cpu().numpy(), single_msgs[k].cpu().numpy(), rtol=1e-11, atol=1e-300)
