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


def random_spec(rs, name, X=4):
    """A connected random graph: a random tree over 3-9 variables with shuffled, non-contiguous ids, up to three extra
    pairwise factors (loops, also parallel ones), unary factors on a random subset, factor ids in random creation
    order, table axes either way round."""
    n = int(rs.randint(3, 10))
    vids = sorted(rs.choice(40, size=n, replace=False).tolist())
    rs.shuffle(vids)
    edges = [(vids[i], vids[int(rs.randint(0, i))]) for i in range(1, n)]
    for _ in range(int(rs.randint(0, 4))):
        a, b = rs.choice(n, size=2, replace=False)
        edges.append((vids[a], vids[b]))
    facs = [[v] for v in vids if rs.rand() < 0.7] + [[a, b] if rs.rand() < 0.5 else [b, a] for a, b in edges]
    order = rs.permutation(len(facs))
    ids = sorted(rs.choice(200, size=len(facs), replace=False).tolist())
    factors = []
    for k, j in enumerate(order):
        vs = facs[j]
        dims = [0] if len(vs) == 1 else ([0, 1] if rs.rand() < 0.5 else [1, 0])
        factors.append(dict(id=int(ids[k]), vars=[int(v) for v in vs], dims=dims, table=k))
    rs.shuffle(factors)                                  # creation order != id order
    seen = []
    for f in factors:
        for v in f['vars']:
            if v not in seen:
                seen.append(v)
    return dict(name=name, style='explicit', X=X, var_ids=seen, labels=[0] * len(seen), factors=factors)


