"""TEST INFRASTRUCTURE -- CPU (NumPy) restatement of the reference's loopy-BP hot path.

This is the *checker*, not the product.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product (`macaronicusermodeling_amd`) runs the
same algorithm in hand-written HIP kernels behind `libmlbp.so` and has no CPU fallback.

Parity pin: `tests/test_oracle_golden.py` checks every function here against fixtures produced by
RUNNING the reference (`tests/golden/*.npz`, generator `tests/golden/make_golden.py`):
schedules and loop tests bit-exact, messages / marginals / beliefs / gradients to 1e-12.

Shape of the restatement (deliberately not the reference's object graph): a graph is the plain
*spec* of `tests/golden/cases.py`; `Graph` flattens it into integer adjacency once, and the
algorithm is a set of functions over a dict of (X,) float64 messages keyed like the reference's
`graph.messages` (`('X_3','F_7')`).  It works one graph at a time with one small NumPy call per
message, i.e. it has the reference's cost model -- which is why bench.py times it as the CPU
baseline (`cpu_baseline.kind = "port"`).

Citations are into /root/reference.
"""
import numpy as np

from . import array_oracle as au

VAR, FAC = 0, 1


class Graph:
    """Integer view of a spec.

    factors are stored in id order (`initialize` sorts them, LBP.py:195-196); a variable's
    neighbour list keeps factor CREATION order (`add_varset_with_potentials` appends, LBP.py:449-452);
    `var_order` is the insertion order of `FactorGraph.variables` (LBP.py:149-153)."""

    def __init__(self, spec):
        self.spec = spec
        self.X = spec['X']
        self.label = dict(zip(spec['var_ids'], spec['labels']))
        self.factors = sorted(spec['factors'], key=lambda f: f['id'])
        self.by_id = {f['id']: f for f in self.factors}
        self.facset = {}
        self.var_order = []
        for f in spec['factors']:                       # creation order
            for v in f['vars']:
                if v not in self.facset:
                    self.facset[v] = []
                    self.var_order.append(v)
                self.facset[v].append(f['id'])

    def neighbours(self, node):
        kind, i = node
        if kind == VAR:
            return [(FAC, f) for f in self.facset[i]]
        return [(VAR, v) for v in self.by_id[i]['vars']]

    def dim_of(self, f, v):
        return f['dims'][f['vars'].index(v)]


def name(node):
    return ('X_%d' if node[0] == VAR else 'F_%d') % node[1]


# ------------------------------------------------------------------------------------------------
# integer work (bit-exact)
# ------------------------------------------------------------------------------------------------
def message_schedule(g, root):
    """LBP.py:155-172 -- breadth-first from the root variable with a FIFO that may hold a node more
    than once; a node is expanded the first time it is dequeued; each not-yet-expanded neighbour
    yields a (child, parent) pair, so a loop-closing factor appears under both of its variables."""
    done, fifo, pairs = [], [(VAR, root)], []
    while fifo:
        n = fifo.pop(0)
        if n in done:
            continue
        done.append(n)
        fresh = [m for m in g.neighbours(n) if m not in done]
        pairs.extend((m, n) for m in fresh)
        fifo.extend(fresh)
    return pairs


def has_loops(g, root):
    """LBP.py:174-190 -- depth-first (LIFO) from the root with the arrival edge excluded; reaching
    an already-visited node means a cycle.  Sees only the root's component."""
    visited, stack = [], [((VAR, root), None)]
    while stack:
        n, came_from = stack.pop()
        if n in visited:
            return True
        visited.append(n)
        stack.extend((m, n) for m in g.neighbours(n) if m != came_from)
    return False


# ------------------------------------------------------------------------------------------------
# tables
# ------------------------------------------------------------------------------------------------
def factor_table(g, inputs, f):
    """explicit style: PotentialTable.__init__ (LBP.py:676-689).  trainmp style: FactorNode.get_pot
    (LBP.py:456-467) then PotentialTable.slice_potentials (LBP.py:695-708)."""
    if g.spec['style'] == 'explicit':
        return np.asarray(inputs['tables'][f['table']], dtype=np.float64)
    if f['factor_type'] == 'en_en':
        if f['gap'] > 1:
            pot = inputs['pot_en_en']
        elif f['gap'] == 1:
            pot = inputs['pot_en_en_w1']
        else:
            raise BaseException('only 2 kinds of distances are supported ...')
    elif f['factor_type'] == 'en_de':
        pot = inputs['pot_en_de']
    else:
        raise BaseException('only two kinds of potentials are supported...')
    if f['observed_dim'] is not None:
        return pot[:, f['observed_dim']].reshape(-1, 1).astype(np.float64)
    return pot.astype(np.float64)


def factor_phi(g, inputs, f):
    """FactorNode.get_phi, LBP.py:469-480."""
    if f['factor_type'] == 'en_en':
        if f['gap'] > 1:
            return inputs['phi_en_en']
        if f['gap'] == 1:
            return inputs['phi_en_en_w1']
        raise BaseException('only 2 distances supported at the moment')
    if f['factor_type'] == 'en_de':
        return inputs['phi_en_de']
    raise BaseException('only 2 feature value types are supported right now..')


# ------------------------------------------------------------------------------------------------
# messages
# ------------------------------------------------------------------------------------------------
def renormalize(m):
    """Message.renormalize, LBP.py:649-657: positive total -> au.normalize; else uniform."""
    if np.sum(m) > 0:
        return au.normalize(m)
    return np.full_like(m, 1.0 / m.size)


def init_messages(g):
    """FactorGraph.initialize, LBP.py:201-216: every message starts uniform; a unary factor has only
    its factor->variable message."""
    msgs = {}
    for f in g.factors:
        fn = 'F_%d' % f['id']
        for v in f['vars']:
            if len(f['vars']) == 2:
                msgs['X_%d' % v, fn] = np.full(g.X, 1.0 / g.X)
            msgs[fn, 'X_%d' % v] = np.full(g.X, 1.0 / g.X)
    return msgs


def _product_of_incoming(g, msgs, v, skip=None):
    """VariableNode.update_message_to / get_marginal, LBP.py:381-386, 393-396: start uniform, fold
    in the incoming factor messages in facset order, `nan_to_num` after every product (LBP.py:728-729)."""
    acc = np.full(g.X, 1.0 / g.X)
    for fid in g.facset[v]:
        if fid != skip:
            acc = np.nan_to_num(au.pointwise_multiply(msgs['F_%d' % fid, 'X_%d' % v], acc))
    return acc


def var_to_factor(g, msgs, v, fid):
    """VariableNode.update_message_to, LBP.py:377-389."""
    msgs['X_%d' % v, 'F_%d' % fid] = renormalize(_product_of_incoming(g, msgs, v, skip=fid))


def factor_to_var(g, inputs, msgs, fid, v, approx=False):
    """FactorNode.update_message_to, LBP.py:490-526."""
    f = g.by_id[fid]
    T = factor_table(g, inputs, f)
    if len(f['vars']) == 1:
        out = np.copy(T).reshape(-1)
    else:
        other = [u for u in f['vars'] if u != v][0]
        m = msgs['X_%d' % other, 'F_%d' % fid].reshape(-1, 1)
        if g.dim_of(f, other) == 1:
            out = au.sparse_vec_mat_dot(m, T) if approx else au.dense_dot(T, m)
        else:
            out = au.sparse_vec_mat_dot(m.T, T) if approx else au.dense_dot(m.T, T)
        out = np.asarray(out).reshape(-1)
    msgs['F_%d' % fid, 'X_%d' % v] = renormalize(out)


def _send(g, inputs, msgs, frm, to, approx):
    if to[0] == FAC and len(g.by_id[to[1]]['vars']) < 2:
        return                                          # LBP.py:228-229, 237-238
    if frm[0] == VAR:
        var_to_factor(g, msgs, frm[1], to[1])
    else:
        factor_to_var(g, inputs, msgs, frm[1], to[1], approx)


def sweep(g, inputs, msgs, root, approx=False):
    """One iteration of FactorGraph.treelike_inference, LBP.py:223-243: leaves->root over the
    reversed schedule (child sends to parent), then root->leaves (parent sends to child)."""
    sched = message_schedule(g, root)
    for child, parent in reversed(sched):
        _send(g, inputs, msgs, child, parent, approx)
    for child, parent in sched:
        _send(g, inputs, msgs, parent, child, approx)


def treelike_inference(g, inputs, msgs, iterations, roots, is_loopy, approx=False):
    """LBP.py:218-245; `roots` replaces the per-sweep `random.sample` draw.  Returns sweeps run."""
    iterations = iterations if is_loopy else 1
    for i in range(iterations):
        sweep(g, inputs, msgs, roots[i], approx)
    return iterations


# ------------------------------------------------------------------------------------------------
# read-outs
# ------------------------------------------------------------------------------------------------
def marginal(g, msgs, v):
    """VariableNode.get_marginal, LBP.py:392-400."""
    return renormalize(_product_of_incoming(g, msgs, v))


def log_posterior(g, msgs):
    """FactorGraph.get_posterior_probs, LBP.py:247-259."""
    total = 0.0
    for v in g.var_order:
        with np.errstate(divide='ignore'):
            lp = np.log(marginal(g, msgs, v)[g.label[v]])
        total += -99.99 if lp == float('-inf') else lp
    return total


def top_indices(p, top):
    """VariableNode.get_max_vocab, LBP.py:402-410: the `top` largest entries, descending."""
    idx = np.argpartition(p, -top)[-top:]
    return idx[np.argsort(p[idx])][::-1]


def precision_counts(g, msgs):
    """FactorGraph.get_precision_counts, LBP.py:80-106."""
    at0 = at25 = at50 = total = 0
    for f in g.factors:
        if f.get('factor_type') != 'en_de':
            continue
        v = f['vars'][0]
        total += 1
        for rank, i in enumerate(top_indices(marginal(g, msgs, v), 50)):
            if i == g.label[v]:
                if rank == 0:
                    at0 += 1; at25 += 1; at50 += 1
                elif rank < 26:
                    at25 += 1; at50 += 1
                elif rank < 51:
                    at50 += 1
    return at0, at25, at50, total


def factor_beliefs(g, inputs, msgs, fid, approx=False):
    """FactorNode.get_factor_beliefs, LBP.py:528-574.  Unary: the normalised table alone.  Pairwise:
    normalise((c r^T) * T), c / r = messages from the dim-0 / dim-1 variable."""
    f = g.by_id[fid]
    T = factor_table(g, inputs, f)
    if len(f['vars']) == 1:
        return au.normalize(np.array(T))
    c = r = None
    for v in f['vars']:
        m = msgs['X_%d' % v, 'F_%d' % fid]
        if g.dim_of(f, v) == 0:
            c = m.reshape(-1, 1)
        else:
            r = m.reshape(1, -1)
    if approx:
        outer, ci, ri = au.sparse_dot(c, r)
        return au.sparse_normalize(au.sparse_pointwise_multiply(outer, ci, ri, T), ci, ri)
    return au.normalize(au.dense_pointwise_multiply(au.dense_dot(c, r), T))


def observed_cell(g, f):
    """FactorNode.get_observed_factor, LBP.py:584-589: label indices ordered by table axis."""
    return tuple(g.label[v] for _, v in sorted((g.dim_of(f, v), v) for v in f['vars']))


def factor_gradient(g, inputs, msgs, fid, approx=False):
    """FactorNode.get_gradient, LBP.py:592-613 with cell_gradient, LBP.py:615-619."""
    f = g.by_id[fid]
    b = factor_beliefs(g, inputs, msgs, fid, approx)
    cell = np.zeros_like(b)
    cell[observed_cell(g, f) if len(f['vars']) == 2 else (g.label[f['vars'][0]], 0)] = 1.0
    cell -= b
    phi = factor_phi(g, inputs, f)
    if f['observed_dim'] is not None:
        grad = np.dot(cell.T, phi[:, f['observed_dim'], :])
    else:
        grad = np.tensordot(cell, phi)
    return grad.reshape(1, -1)


def unregularized_gradient(g, inputs, msgs, approx=False):
    """FactorGraph.get_unregularized_gradeint, LBP.py:301-320 -> (en_en, en_de)."""
    g_ee = np.zeros_like(inputs['theta_en_en'], dtype=np.float64)
    g_ed = np.zeros_like(inputs['theta_en_de'], dtype=np.float64)
    for f in g.factors:
        if f['factor_type'] == 'en_en':
            g_ee += factor_gradient(g, inputs, msgs, f['id'], approx)
        elif f['factor_type'] == 'en_de':
            g_ed += factor_gradient(g, inputs, msgs, f['id'], approx)
        else:
            raise BaseException('only 2 kinds of factors allowed...')
    return g_ee, g_ed


def regularized_gradient(g, inputs, msgs, reg, approx=False):
    """FactorGraph.get_gradient, LBP.py:293-299 -> (en_de, en_en)  [note the order]."""
    g_ee, g_ed = unregularized_gradient(g, inputs, msgs, approx)
    return g_ed - reg * inputs['theta_en_de'], g_ee - reg * inputs['theta_en_en']


def return_gradient(g, inputs, msgs, reg, lr, approx=False):
    """FactorGraph.return_gradient, LBP.py:322-327 -> (en_en, en_de) scaled by the learning rate."""
    g_ed, g_ee = regularized_gradient(g, inputs, msgs, reg, approx)
    return lr * g_ee, lr * g_ed


# ------------------------------------------------------------------------------------------------
# convenience used by tests / smoke / cpu_baseline
# ------------------------------------------------------------------------------------------------
def run(spec, inputs, roots, sweeps, force_loopy=False, approx=False):
    """initialize + treelike_inference on one graph; returns (Graph, messages, sweeps_run)."""
    g = Graph(spec)
    loopy = True if force_loopy else has_loops(g, roots[0])
    msgs = init_messages(g)
    n = treelike_inference(g, inputs, msgs, sweeps, roots, loopy, approx)
    return g, msgs, n
