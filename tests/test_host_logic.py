"""Host-side integer logic of libmlbp.so (C++), bit-exact against reference-generated fixtures.
Runs without a GPU."""
import numpy as np
import pytest

import cases as C
from conftest import load_golden
from macaronicusermodeling_amd import _ffi
from macaronicusermodeling_amd.topology import GraphTopology
from helpers import random_spec as _random_spec
from oracle import lbp_oracle as O


def _encode(topo, pairs):
    out = []
    for a, b in pairs:
        row = []
        for n in (a, b):
            name = topo.node_name(n)
            row += list(C.node_code(name))
        out.append(row)
    return np.array(out, dtype=np.int64).reshape(-1, 4)


def test_schedules_and_loop_test_match_reference_fixtures():
    gold = load_golden('schedules')
    n = 0
    for spec in C.schedule_topologies():
        topo = GraphTopology.from_spec(spec)
        for vid in topo.var_ids:
            assert topo.has_loops(vid) == bool(gold['%s/loops_root%d' % (spec['name'], vid)])
            np.testing.assert_array_equal(_encode(topo, topo.message_schedule(vid)),
                                          gold['%s/sched_root%d' % (spec['name'], vid)])
            n += 1
    assert n == len([k for k in gold.files if '/sched_' in k])


@pytest.mark.parametrize('case', C.inference_cases(), ids=lambda c: c['name'])
def test_case_schedules(case):
    gold = load_golden(case['name'])
    topo = GraphTopology.from_spec(case['spec'])
    assert topo.has_loops(case['roots'][0]) == bool(gold['is_loopy'])
    for r in sorted(set(case['roots'])):
        np.testing.assert_array_equal(_encode(topo, topo.message_schedule(r)), gold['sched_root%d' % r])
    assert topo.slot_keys() == C.msg_keys(case['spec'])
    np.testing.assert_array_equal(np.array(topo.var_ids), gold['var_order'])


def _oracle_ops(spec, root):
    """The op list implied by the oracle's own schedule walk, for comparison with the compiler."""
    g = O.Graph(spec)
    topo = GraphTopology.from_spec(spec)
    slot = {k: i for i, k in enumerate(C.msg_keys(spec))}
    ops = []
    sched = O.message_schedule(g, root)
    sends = [(c, p) for c, p in reversed(sched)] + [(p, c) for c, p in sched]
    pair_ids = [f['id'] for f in g.factors if len(f['vars']) == 2]
    unary_ids = [f['id'] for f in g.factors if len(f['vars']) == 1]
    for frm, to in sends:
        if to[0] == O.FAC and len(g.by_id[to[1]]['vars']) < 2:
            continue
        if frm[0] == O.VAR:
            srcs = [slot['F_%d' % f, 'X_%d' % frm[1]] for f in g.facset[frm[1]] if f != to[1]]
            ops.append(('var', tuple(srcs), slot['X_%d' % frm[1], 'F_%d' % to[1]]))
        else:
            f = g.by_id[frm[1]]
            dst = slot['F_%d' % f['id'], 'X_%d' % to[1]]
            if len(f['vars']) == 1:
                ops.append(('unary', unary_ids.index(f['id']), dst))
            else:
                other = [u for u in f['vars'] if u != to[1]][0]
                kind = 'tm' if g.dim_of(f, other) == 1 else 'mt'
                ops.append((kind, pair_ids.index(f['id']), slot['X_%d' % other, 'F_%d' % f['id']], dst))
    return ops


@pytest.mark.parametrize('spec', C.schedule_topologies() + [c['spec'] for c in C.inference_cases()[:9]],
                         ids=lambda s: s['name'])
def test_compiled_sweep_equals_oracle_walk(spec):
    topo = GraphTopology.from_spec(spec)
    for root in topo.var_ids:
        ops, srcs = topo.compile_sweep(root)
        got = []
        for kind, a, b, c in ops.tolist():
            if kind == _ffi.OP_VAR:
                got.append(('var', tuple(srcs[a:a + b].tolist()), c))
            elif kind == _ffi.OP_UNARY:
                got.append(('unary', a, c))
            else:
                got.append(('tm' if kind == _ffi.OP_PAIR_TM else 'mt', a, b, c))
        assert got == _oracle_ops(spec, root)


def test_update_counts_per_sweep():
    """per sweep exactly 2P pairwise f->v, U unary f->v, 2P v->f."""
    for spec in (C.chain_spec(8, 4), C.ring_spec(8, 4), C.user_spec(10, [1, 4, 7], 4, 4)):
        topo = GraphTopology.from_spec(spec)
        ops, _ = topo.compile_sweep(topo.var_ids[0])
        kinds = ops[:, 0]
        assert ((kinds == _ffi.OP_PAIR_TM) | (kinds == _ffi.OP_PAIR_MT)).sum() == 2 * topo.P
        assert (kinds == _ffi.OP_UNARY).sum() == topo.U
        assert (kinds == _ffi.OP_VAR).sum() == 2 * topo.P


def test_program_concatenation_shares_equal_roots():
    topo = GraphTopology.from_spec(C.ring_spec(8, 4))
    ops, srcs, sweeps = topo.compile_program([0, 3, 0, 0])
    one, _ = topo.compile_sweep(0)
    assert len(ops) == 2 * len(one)
    assert sweeps.tolist() == [[0, len(one)], [len(one), len(one)], [0, len(one)], [0, len(one)]]


def test_bad_topologies_are_rejected():
    with pytest.raises(NotImplementedError):
        GraphTopology([(0, [0, 1, 2], [0, 1, 2])])
    with pytest.raises(_ffi.MlbpError):
        GraphTopology([(0, [0, 1], [0, 0])])
    with pytest.raises(_ffi.MlbpError):
        GraphTopology([(0, [0, 0], [0, 1])])
    with pytest.raises(ValueError):
        GraphTopology([(0, [0], [0]), (0, [1], [0])])


def test_program_rewrites_plan():
    """Host-side program rewrites (no GPU): after sinking variable->factor updates next to their consumers every
    update inside a K3 sweep fuses (what is left lone / standalone are the sweep's last two variable updates and
    its first two pairwise updates, which straddle the sweep boundary); with a different root per sweep the last two
    variable updates of sweeps 1 and 2 are overwritten before anything reads them and are dropped (24 -> 20 updates),
    with the same root every sweep they feed the next sweep and stay; the shared-table form of the same call
    keeps 9 message tiles resident (3 constant products + 6 factor->variable messages) = 78 336 bytes for 16
    graphs, i.e. two workgroups per CU; K4 needs 21 tiles (the launcher keeps 16 in LDS and spills the 5 stored
    variable->factor messages to global memory)."""
    from macaronicusermodeling_amd.topology import GraphTopology
    k3 = GraphTopology.from_spec(C.user_spec(10, [1, 4, 7], 64, 64, seed=1)).plan([1, 4, 7])
    assert (k3['updates'], k3['fused_updates'], k3['lone_variable_updates'], k3['bundles']) == (20, 12, 2, 9)
    same = GraphTopology.from_spec(C.user_spec(10, [1, 4, 7], 64, 64, seed=1)).plan([1, 1, 1])
    assert (same['updates'], same['lone_variable_updates']) == (24, 6)
    assert (k3['shared_ok'], k3['shared_tiles'], k3['shared_tile_bytes']) == (1, 9, 78336)
    assert 2 * k3['shared_tile_bytes'] < 160 * 1024
    k2 = GraphTopology.from_spec(C.user_spec(6, [0, 1], 64, 64, seed=3)).plan([0, 1, 0])
    assert (k2['updates'], k2['fused_updates'], k2['lone_variable_updates'], k2['shared_tiles']) == (6, 6, 0, 4)
    k4 = GraphTopology.from_spec(C.user_spec(9, [0, 2, 3, 7], 64, 64, seed=4)).plan([0, 2, 3])
    assert k4['shared_tiles'] == 21 and k4['shared_tile_bytes'] > 160 * 1024
    # the product-fused form of the shared-table kernel: variables with at most two pairwise factors (K2, K3); K3's final
    # variable->factor messages are message tiles at the end of the sweeps (the gradient epilogue reads them there), K2's are
    # the constant products themselves; K4's variable updates multiply two messages: the general form
    assert (k3['shared_product_fused'], k3['shared_gradient_from_tiles']) == (1, 1)
    assert (k2['shared_product_fused'], k2['shared_gradient_from_tiles']) == (1, 0)
    assert (k4['shared_product_fused'], k4['shared_gradient_from_tiles']) == (0, 0)
    # K4: the three-source variant (messages stored as sqrt(c) (.) message, 12 + 5 tiles in LDS, the constant products in memory)
    assert (k4['shared_product_fused3'], k3['shared_product_fused3'], k2['shared_product_fused3']) == (1, 0, 0)
    ring = GraphTopology.from_spec(C.ring_spec(8, 64)).plan([0, 0, 0])
    assert ring['fused_updates'] == ring['updates'] == 48 and ring['bundles'] == 24
    k1 = GraphTopology.from_spec(C.user_spec(5, [2], 64, 64, seed=5)).plan([2])      # no pairwise factor at all
    assert k1['shared_ok'] == 0 and k1['updates'] == 0


@pytest.mark.parametrize('seed', range(25))
def test_random_graphs_schedule_loop_test_and_compiled_sweeps_equal_the_oracle_walk(seed):
    """The C++ host logic against the oracle's restatement of LBP.py:155-190, 223-243 on random graphs with shuffled
    ids, creation orders and table axes (the fixtures pin the oracle on eleven hand-made shapes; this carries it to shapes
    nobody drew)."""
    rs = np.random.RandomState(9000 + seed)
    spec = _random_spec(rs, 'random_%d' % seed)
    g = O.Graph(spec)
    topo = GraphTopology.from_spec(spec)
    assert topo.slot_keys() == C.msg_keys(spec)
    for root in topo.var_ids:
        assert topo.has_loops(root) == O.has_loops(g, root)
        want = [(O.name(a), O.name(b)) for a, b in O.message_schedule(g, root)]
        got = [(topo.node_name(a), topo.node_name(b)) for a, b in topo.message_schedule(root)]
        assert got == want
        ops, srcs = topo.compile_sweep(root)
        mine = []
        for kind, a, b, c in ops.tolist():
            if kind == _ffi.OP_VAR:
                mine.append(('var', tuple(srcs[a:a + b].tolist()), c))
            elif kind == _ffi.OP_UNARY:
                mine.append(('unary', a, c))
            else:
                mine.append(('tm' if kind == _ffi.OP_PAIR_TM else 'mt', a, b, c))
        assert mine == _oracle_ops(spec, root)
