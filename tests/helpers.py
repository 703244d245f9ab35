"""Test helpers: turn case specs + seeded inputs into FactorGraphBatch device tensors."""
import numpy as np

import cases as C
from oracle import lbp_oracle as O


def batch_tables(spec, topo, inputs_list):
    """pair tables [B*P][X][X] and unary tables [B*U][X] in (graph, slot) order."""
    g = O.Graph(spec)
    X = spec['X']
    pair, unary = [], []
    for inputs in inputs_list:
        for j in topo.pair_factors:
            pair.append(O.factor_table(g, inputs, g.by_id[topo.factor_ids[j]]).reshape(X, X))
        for j in topo.unary_factors:
            unary.append(O.factor_table(g, inputs, g.by_id[topo.factor_ids[j]]).reshape(X))
    pair = np.stack(pair) if pair else np.zeros((0, X, X))
    unary = np.stack(unary) if unary else np.zeros((0, X))
    return pair, unary


def case_inputs(case, n_graphs):
    """Graph 0 = the fixture's inputs; further graphs use other seeds."""
    kind = case['kind'] or 'uniform'
    return [C.make_inputs(case['spec'], case['seed'] + 1000 * i, kind) for i in range(n_graphs)]


def oracle_msgs(spec, inputs, roots, force_loopy=True):
    g = O.Graph(spec)
    msgs = O.init_messages(g)
    for r in roots:
        O.sweep(g, inputs, msgs, r)
    keys = C.msg_keys(spec)
    return g, msgs, np.stack([msgs[k] for k in keys])
