"""MI355X-backed drop-in for the reference module `LBP.py`.

    from macaronicusermodeling_amd.LBP import (FactorNode, FactorGraph, VariableNode, PotentialTable,
                                                VAR_TYPE_PREDICTED, VAR_TYPE_GIVEN, PhiWrapper)

Same classes, constructor signatures, attributes, method names, return shapes and error behaviour
as the reference (cited per method as LBP.py:<lines> of the reference).  What differs is where the
arithmetic happens: the object graph is host bookkeeping only; every message update, product,
normalisation, marginal, belief and gradient contraction is a HIP kernel behind libmlbp.so
(include/mlbp.h).  `initialize()` flattens the graph into a `GraphTopology` + a one-graph
`FactorGraphBatch`; `treelike_inference()` compiles the drawn root sequence into an op list and
runs ALL sweeps in one fused launch; `graph.messages` is a lazy dict view over the device tensor.
There is no CPU fallback.
"""
import ctypes as C
import hashlib
import random
import sys
import time

import numpy as np
import torch
from numpy import float64 as DTYPE

from . import _ffi
from .array_utils import c_array_utils as au
from .batch import FactorGraphBatch, Program, _stream_ptr
from .topology import GraphTopology

VAR_TYPE_PREDICTED = 'var_type_predicted'
VAR_TYPE_GIVEN = 'var_type_given'
VAR_TYPE_LATENT = 'var_type_latent'
UNARY_FACTOR = 'unary_factor'
BINARY_FACTOR = 'binary_factor'


def _device():
    if not torch.cuda.is_available():
        raise _ffi.MlbpError(_ffi.MLBP_ENODEVICE, 'no MI355X visible: LBP has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _dev_log(values):
    """np.log of a small host array, evaluated on the device (mlbp_log_f64)."""
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(dev)
    out = torch.empty_like(t)
    _ffi.check(_ffi.lib.mlbp_log_f64(t.data_ptr(), out.data_ptr(), t.numel(), _stream_ptr(dev)))
    return out.cpu().numpy().reshape(np.shape(values))


# ------------------------------------------------------------------------------------------------------
# device engine behind one FactorGraph
# ------------------------------------------------------------------------------------------------------
class _Engine:
    """Flattened form of an initialised FactorGraph: topology, one-graph batch, message mirror."""

    def __init__(self, fg):
        self.fg = fg
        factors = fg.factors
        known = set(f.id for f in factors)
        for v in fg.variables.values():
            for f in v.facset:
                if f.id not in known:
                    raise KeyError((str(f), str(v)))       # the reference fails on the message lookup
        sizes = set(len(v.domain) for v in fg.variables.values())
        if len(sizes) != 1:
            raise NotImplementedError('all variables of a graph must share one domain size (square pairwise '
                                      'tables, LBP.py:691)')
        self.X = sizes.pop()
        self.topo = GraphTopology(
            [(f.id, [v.id for v in f.varset], [f.potential_table.var_id2dim[v.id] for v in f.varset]) for f in factors],
            var_ids=list(fg.variables.keys()),
            facsets={vid: [f.id for f in v.facset] for vid, v in fg.variables.items()})
        self.batch = FactorGraphBatch(self.topo, self.X, 1, device=_device(), share_programs=True)
        self.keys = self.topo.slot_keys()
        self.slot = {k: i for i, k in enumerate(self.keys)}
        self.host = np.full((self.topo.n_msgs, self.X), 1.0 / self.X)
        self.host_fresh = True
        self.dev_fresh = False
        self._one_op = {}
        self._table_sig = None

    # ---- message mirror -------------------------------------------------------------------------
    def to_host(self):
        if not self.host_fresh:
            self.host = self.batch.msgs[0].cpu().numpy()
            self.host_fresh = True
        return self.host

    def to_device(self):
        if not self.dev_fresh:
            self.batch.msgs[0].copy_(torch.from_numpy(np.ascontiguousarray(self.host)))
            self.dev_fresh = True

    def device_changed(self):
        self.host_fresh = False
        self.dev_fresh = True

    # ---- tables: the reference reads PotentialTable.table afresh on every update, so edits made between two calls
    #      must be seen -- but re-sending every table for one message update is O(P X^2) bytes over PCIe.  The device
    #      copies are kept; each call fingerprints the host arrays (shape and a 128-bit BLAKE2 digest of the bytes) and re-sends
    #      only the tables that changed.  An unchanged graph moves no table bytes at all.
    @staticmethod
    def _fingerprint(arr):
        # content only: an in-place edit keeps the address, and a converted copy's address means nothing
        return (arr.shape, hashlib.blake2b(arr, digest_size=16).digest())

    def upload_tables(self):
        t, X = self.topo, self.X
        by_index = {self.topo.factor_index[f.id]: f for f in self.fg.factors}
        pair = [np.ascontiguousarray(by_index[j].potential_table.table, dtype=np.float64).reshape(X, X) for j in t.pair_factors]
        un = [np.ascontiguousarray(by_index[j].potential_table.table, dtype=np.float64).reshape(X) for j in t.unary_factors]
        sig = [self._fingerprint(a) for a in pair + un]
        if self._table_sig is None:
            if t.P:
                self.batch.set_pair_tables(np.stack(pair))
            if t.U:
                self.batch.set_unary_tables(np.stack(un))
        else:
            for k, a in enumerate(pair):
                if sig[k] != self._table_sig[k]:
                    self.batch.pair_tables[k].copy_(torch.from_numpy(a))
            for k, a in enumerate(un):
                if sig[t.P + k] != self._table_sig[t.P + k]:
                    self.batch.unary_tables[k].copy_(torch.from_numpy(a))
        self._table_sig = sig
        self.batch.normalize_messages = bool(self.fg.normalize_messages)

    def run_sweeps(self, roots):
        self.upload_tables()
        self.to_device()
        self.batch.sweep(roots)
        self.device_changed()

    def run_op(self, op, srcs=()):
        """One message update as a one-op program (VariableNode / FactorNode.update_message_to)."""
        key = (tuple(op), tuple(srcs))
        if key not in self._one_op:
            prog = Program.__new__(Program)
            ops = np.array([op], dtype=np.int32)
            sr = np.array(list(srcs) or [0], dtype=np.int32)
            sw = np.array([0, 1], dtype=np.int32)
            h = C.c_void_p()
            _ffi.check(_ffi.lib.mlbp_program_create(_ffi.i32ptr(ops.reshape(-1)), 1, _ffi.i32ptr(sr), len(srcs),
                                                    _ffi.i32ptr(sw), 1, self.topo.n_msgs, self.topo.P, self.topo.U,
                                                    C.byref(h)))
            prog.handle, prog.roots = h, None
            self._one_op[key] = prog
        self.upload_tables()
        self.to_device()
        b = self.batch
        a = _ffi.SweepArgs()
        a.B, a.X = 1, self.X
        if self.topo.P:
            a.n_pair_tables, a.pair_tables, a.pair_tab = b.pair_tables.shape[0], b.pair_tables.data_ptr(), b.pair_tab.data_ptr()
        if self.topo.U:
            a.n_unary_tables, a.unary_tables, a.unary_tab = b.unary_tables.shape[0], b.unary_tables.data_ptr(), b.unary_tab.data_ptr()
        a.msgs = b.msgs.data_ptr()
        a.normalize_messages = 1 if self.fg.normalize_messages else 0
        _ffi.check(_ffi.lib.mlbp_sweep_f64(self._one_op[key].handle, C.byref(a), _stream_ptr(b.device)))
        self.device_changed()

    def marginals(self):
        self.to_device()
        self.batch.normalize_messages = bool(self.fg.normalize_messages)
        return self.batch.marginals()


class _MessageStore:
    """`FactorGraph.messages`: dict keyed (str(src), str(dst)) -> Message, backed by the device."""

    def __init__(self, engine):
        self._e = engine

    def __getitem__(self, key):
        """A COPY of the message (the values live on the device): unlike the reference's dict, editing `messages[k].m` in
        place changes nothing -- assign the Message back (`messages[k] = msg`) to write it."""
        i = self._e.slot[key]                         # KeyError like a dict
        return Message(self._e.to_host()[i].copy())

    def __setitem__(self, key, msg):
        i = self._e.slot[key]
        m = msg.m if isinstance(msg, Message) else np.asarray(msg)
        host = self._e.to_host()
        host[i] = np.asarray(m, dtype=np.float64).reshape(-1)
        self._e.dev_fresh = False

    def __contains__(self, key):
        return key in self._e.slot

    def __len__(self):
        return len(self._e.keys)

    def __iter__(self):
        return iter(self._e.keys)

    def keys(self):
        return list(self._e.keys)

    def values(self):
        return [self[k] for k in self._e.keys]

    def items(self):
        return [(k, self[k]) for k in self._e.keys]


# ------------------------------------------------------------------------------------------------------
class FactorGraph():
    """LBP.py:19-333."""

    def __init__(self,
                 theta_en_en_names,
                 theta_en_de_names,
                 theta_en_en,
                 theta_en_de,
                 phi_en_en_w1,
                 phi_en_en,
                 phi_en_de):
        # caller-visible state, same names and defaults as LBP.py:28-57 (the reference's accidental
        # 1-tuples around the *_names arguments are unwrapped there too, so plain lists are kept)
        self.__dict__.update(
            theta_en_en=theta_en_en, theta_en_de=theta_en_de,
            theta_en_en_names=theta_en_en_names, theta_en_de_names=theta_en_de_names,
            phi_en_en=phi_en_en, phi_en_en_w1=phi_en_en_w1, phi_en_de=phi_en_de,
            pot_en_en=None, pot_en_en_w1=None, pot_en_de=None,
            variables={}, factors=[], messages={}, active_domains={},
            normalize_messages=True, isLoopy=None, regularization_param=0.01, learning_rate=0.1,
            report_times=False, use_approx_inference=False, use_approx_beliefs=False)
        for timer in ('bb_times', 'ub_times', 'it_times', 'gg_times', 'sgg_times'):
            setattr(self, timer, [])
        self._engine = None

    def display_timing_info(self):
        """LBP.py:59-77."""
        if self.report_times:
            for label, ts in (('\nubtimes    :', self.ub_times), ('bbtimes    :', self.bb_times),
                              ('ggtimes    :', self.gg_times), ('sggtimes    :', self.sgg_times),
                              ('it_times    :', self.it_times)):
                if len(ts) > 0:
                    print(label, np.sum(ts) / len(ts), 'total', np.sum(ts), 'len', len(ts))
                elif label.startswith('bbtimes'):
                    print(label, 0, 'total', 0, 'len', 0)
            print('num vars   :', len(self.variables))
        return True

    def get_precision_counts(self):
        """LBP.py:80-106."""
        hits = [0, 0, 0]                                   # rank 0 / rank < 26 / rank < 51
        totals = 0
        for f in (f for f in self.factors if f.factor_type == 'en_de'):
            label, _, ranked = f.varset[0].get_max_vocab(50)
            totals += 1
            for rank, (word, _) in enumerate(ranked):      # every occurrence counts, as in the reference
                if word == label:
                    for slot, limit in enumerate((1, 26, 51)):
                        hits[slot] += rank < limit
        return hits[0], hits[1], hits[2], totals

    def to_string(self):
        """LBP.py:109-123."""
        lines = {}
        for f in self._positioned():
            if f.factor_type == 'en_de':
                label, label_lp, ranked = f.varset[0].get_max_vocab(50)
                lines[f.position] = ' '.join([f.word_label, label, label_lp] + [w + ' ' + lp for w, lp in ranked])
            if f.factor_type == 'en_en':
                lines[f.position] = ' ' + f.word_label + ' '
        return [lines[k] for k in sorted(lines)]

    def _positioned(self):
        """Factors that carry a sentence position, by (position, id) -- Python 3 cannot order
        FactorNode objects the way LBP.py:110,127 relies on."""
        return [f for _, _, f in sorted((f.position, f.id, f) for f in self.factors if f.position is not None)]

    def to_dist(self):
        """LBP.py:125-143: `truth ||| guess ||| log-marginals (%0.6f)` per en_de factor."""
        rows = []
        for f in self._positioned():
            if f.factor_type != 'en_de':
                continue
            v = f.varset[0]
            logs = _dev_log(v.get_marginal().m).reshape(-1)
            rows.append(' ||| '.join([v.truth_label if v.truth_label is not None else 'None',
                                      v.supervised_label if v.supervised_label is not None else 'None',
                                      ' '.join('%0.6f' % x for x in logs)]))
        return '\n'.join(rows)

    def add_factor(self, fac):
        """LBP.py:145-153: the factor joins the list; variables it brings are registered by id, first come first kept."""
        if __debug__:
            assert fac not in self.factors
        fac.graph = self
        self.factors.append(fac)
        newcomers = [v for v in fac.varset if v.id not in self.variables]
        for v in newcomers:
            v.graph = self
            self.variables.setdefault(v.id, v)

    def _topology(self):
        if self._engine is None:
            raise RuntimeError('initialize() first')
        return self._engine.topo

    def get_message_schedule(self, root):
        """LBP.py:155-172 -> [(child node, parent node)] of VariableNode / FactorNode objects."""
        if __debug__: assert isinstance(root, VariableNode)
        topo = self._engine.topo if self._engine is not None else _Engine(self).topo
        f_by_index = {topo.factor_index[f.id]: f for f in self.factors}

        def node(code):
            return self.variables[topo.var_ids[code]] if code < topo.n_vars else f_by_index[code - topo.n_vars]
        return [(node(a), node(b)) for a, b in topo.message_schedule(root.id)]

    def has_loops(self):
        """LBP.py:174-190 (consumes one `random.sample` draw for the root, like the reference)."""
        _rand_key = random.sample(list(self.variables.keys()), 1)[0]
        topo = self._engine.topo if self._engine is not None else _Engine(self).topo
        return topo.has_loops(_rand_key)

    def initialize(self):
        """LBP.py:192-216: sort factors by id, loop test, all messages uniform -- on the device."""
        if __debug__: assert len(self.variables) > 0
        if __debug__: assert len(self.factors) > 0
        self.factors = [f for fid, f in sorted([(f.id, f) for f in self.factors], key=lambda t: t[0])]
        for f in self.factors:
            if __debug__: assert len(f.potential_table.var_id2dim) == len(f.varset)
            vs = [self.variables[vid] for d, vid in sorted([(d, v) for v, d in f.potential_table.var_id2dim.items()])]
            if len(vs) == 2:
                if __debug__: assert np.shape(f.potential_table.table) == tuple([len(v.domain) for v in vs])
            else:
                if __debug__: assert np.shape(f.potential_table.table) == (len(vs[0].domain), 1)
        self._engine = _Engine(self)
        self.isLoopy = self.has_loops()
        self._engine.batch.initialize()
        self._engine.device_changed()
        self.messages = _MessageStore(self._engine)

    def treelike_inference(self, iterations):
        """LBP.py:218-245.  Exact mode: the whole root sequence runs as ONE fused device launch
        (one per sweep when `report_times` wants per-sweep times).  Approximate mode walks the
        schedule update by update like the reference (each update is still device work)."""
        iterations = iterations if self.isLoopy else 1
        roots = [random.sample(list(self.variables.keys()), 1)[0] for _ in range(iterations)]
        if self.use_approx_inference:
            for r in roots:
                if self.report_times: it = time.time()
                _schedule = self.get_message_schedule(self.variables[r])
                for frm, to in reversed(_schedule):
                    if not (isinstance(to, FactorNode) and len(to.varset) < 2):
                        frm.update_message_to(to)
                for to, frm in _schedule:
                    if not (isinstance(to, FactorNode) and len(to.varset) < 2):
                        frm.update_message_to(to)
                if self.report_times: self.it_times.append(time.time() - it)
        elif self.report_times:
            for r in roots:
                it = time.time()
                self._engine.run_sweeps([r])
                torch.cuda.synchronize()
                self.it_times.append(time.time() - it)
        else:
            self._engine.run_sweeps(roots)
        return True

    def get_posterior_probs(self):
        """LBP.py:247-259 (marginals + log + sum on the device; -inf -> -99.99)."""
        e = self._engine
        labels = np.array([[self.variables[v].supervised_label_index for v in e.topo.var_ids]])
        marg = e.marginals()
        lp = e.batch.log_posterior(labels, marginals=marg)
        return float(lp.item())

    def get_max_postior_label(self, top=10):
        """LBP.py:261-267."""
        label_guesses = []
        for v_key, v in self.variables.items():
            s, sp, g = v.get_max_vocab(top)
            label_guesses.append(s + ' ' + sp + ' ' + ' '.join([i + ' ' + p for i, p in g]))
        return label_guesses

    def hw_inf(self, iterations):
        """LBP.py:269-270."""
        raise BaseException("This method assumes self.variables is a list.. depricated...")

    def get_gradient(self):
        """LBP.py:293-299 -> (grad_en_de, grad_en_en), L2-regularised."""
        grad_en_en, grad_en_de = self.get_unregularized_gradeint()
        grad_en_en -= self.regularization_param * self.theta_en_en
        grad_en_de -= self.regularization_param * self.theta_en_de
        return grad_en_de, grad_en_en

    def get_unregularized_gradeint(self):
        """LBP.py:301-320 -> (grad_en_en, grad_en_de)."""
        total = {'en_en': np.zeros_like(self.theta_en_en, dtype=DTYPE), 'en_de': np.zeros_like(self.theta_en_de, dtype=DTYPE)}
        for f in self.factors:
            if f.factor_type not in total:
                raise BaseException('only 2 kinds of factors allowed...')
            total[f.factor_type] += f.get_gradient()
        return total['en_en'], total['en_de']

    def return_gradient(self):
        """LBP.py:322-327 -> (g_en_en, g_en_de) scaled by the learning rate."""
        grad_en_de, grad_en_en = self.get_gradient()
        return self.learning_rate * grad_en_en, self.learning_rate * grad_en_de

    def update_theta(self):
        """LBP.py:329-333."""
        step_en_en, step_en_de = self.return_gradient()          # learning rate x regularised gradient
        np.add(self.theta_en_en, step_en_en, out=self.theta_en_en)      # in place: callers hold references to theta
        np.add(self.theta_en_de, step_en_de, out=self.theta_en_de)
        return self.theta_en_en, self.theta_en_de


class VariableNode():
    """LBP.py:336-411."""

    def __init__(self, id, var_type, domain_type, domain, supervised_label):
        # LBP.py:337-352: a non-integer id is only reported; a supervised label outside the domain ends the process
        # with status -1 (the messages and the exit status are part of the behaviour callers see)
        if not isinstance(id, int):
            print('id ', id, 'not an int')
        try:
            label_index = domain.index(supervised_label)
        except ValueError:
            print(supervised_label, 'not in', domain)
            sys.exit(-1)
        self.id, self.var_type, self.domain_type, self.domain = id, var_type, domain_type, domain
        self.supervised_label, self.supervised_label_index = supervised_label, label_index
        self.truth_label = self.truth_label_index = None
        self.facset, self.graph = [], None

    def set_truth_label(self, tl):
        self.truth_label = tl

    def __str__(self):
        return "X_" + str(self.id)

    def __eq__(self, other):
        return isinstance(other, VariableNode) and self.id == other.id

    __hash__ = object.__hash__

    def display(self, m):
        raise NotImplementedError()

    def add_factor(self, fc):
        if __debug__: assert isinstance(fc, FactorNode)
        self.facset.append(fc)

    def init_message_to(self, fc, init_m):
        if __debug__: assert isinstance(fc, FactorNode)
        if __debug__: assert isinstance(init_m, Message)
        self.graph.messages[str(self), str(fc)] = init_m

    def update_message_to(self, fc):
        """LBP.py:377-389: uniform x product of the other factors' messages, nan_to_num after each
        product, renormalised -- one MLBP_OP_VAR on the device."""
        if __debug__: assert isinstance(fc, FactorNode)
        if __debug__: assert fc in self.facset
        e = self.graph._engine
        srcs = [e.slot[str(o), str(self)] for o in self.facset if o is not fc]
        e.run_op((_ffi.OP_VAR, 0, len(srcs), e.slot[str(self), str(fc)]), srcs)

    def get_marginal(self):
        """LBP.py:392-400."""
        e = self.graph._engine
        row = e.marginals()[0, e.topo.var_index[self.id]].cpu().numpy()
        return Message(row)

    def get_max_vocab(self, top):
        """LBP.py:402-411: the `top` most probable words, descending, with '%0.4f' log-probabilities.
        Selection runs on the device (mlbp_topk_f64; ties -> lower index, the reference leaves tie
        order to np.argpartition)."""
        m = self.get_marginal()
        a = np.reshape(m.m, (np.size(m.m),))
        if top > a.size:
            raise ValueError('kth(=%d) out of bounds (%d)' % (a.size - top, a.size))
        dev = _device()
        t = torch.from_numpy(a).to(dev)
        idx = torch.empty(top, dtype=torch.int32, device=dev)
        _ffi.check(_ffi.lib.mlbp_topk_f64(t.data_ptr(), 1, a.size, top, idx.data_ptr(), _stream_ptr(dev)))
        max_idx = idx.cpu().numpy()
        al = _dev_log(a)
        max_vocab = [(self.domain[i], '%0.4f' % al[i]) for i in max_idx]
        return self.supervised_label, '%0.4f' % al[self.supervised_label_index], max_vocab


class FactorNode():
    """LBP.py:414-628."""

    def __init__(self, id, factor_type=None, observed_domain_type=None, observed_value=None, observed_domain_size=None):
        if __debug__: assert isinstance(id, int)
        self.__dict__.update(id=id, varset=[], potential_table=None, factor_type=factor_type, graph=None,
                             observed_domain_type=observed_domain_type, observed_value=observed_value,
                             observed_domain_size=observed_domain_size,
                             position=None, word_label=None, gap=None, connect_type=None)   # set by callers

    def __str__(self):
        return 'F_' + str(self.id)

    def __eq__(self, other):
        return isinstance(other, FactorNode) and self.id == other.id

    __hash__ = object.__hash__

    def init_message_to(self, var, init_m):
        if __debug__: assert isinstance(var, VariableNode)
        if __debug__: assert isinstance(init_m, Message)
        self.graph.messages[str(self), str(var)] = init_m

    def add_varset_with_potentials(self, varset, ptable):
        """LBP.py:441-454."""
        arity = len(varset)
        if __debug__:
            assert isinstance(ptable, PotentialTable) and arity == len(ptable.var_id2dim)
            assert arity != 2 or varset[0] != varset[1]
        if arity > 2:
            raise NotImplementedError("Currently supporting unary and pairwise factors...")
        for v in varset:
            if __debug__:
                assert v not in self.varset
            v.add_factor(self)
            self.varset.append(v)
        self.potential_table = ptable
        ptable.add_factor(self)

    def _graph_level(self, prefix, distance_error, type_error):
        """Which graph-level array this factor uses: en_en by word distance (gap > 1 / gap == 1),
        en_de otherwise; anything else raises like LBP.py:456-480."""
        if self.factor_type == 'en_en':
            if self.gap > 1:
                return getattr(self.graph, prefix + '_en_en')
            if self.gap == 1:
                return getattr(self.graph, prefix + '_en_en_w1')
            raise BaseException(distance_error)
        if self.factor_type == 'en_de':
            return getattr(self.graph, prefix + '_en_de')
        raise BaseException(type_error)

    def get_pot(self):
        """LBP.py:456-467."""
        return self._graph_level('pot', "only 2 kinds of distances are supported ...",
                                 "only two kinds of potentials are supported...")

    def get_phi(self):
        """LBP.py:469-480."""
        return self._graph_level('phi', "only 2 distances supported at the moment",
                                 "only 2 feature value types are supported right now..")

    def get_shape(self):
        """LBP.py:482-488."""
        if len(self.varset) not in (1, 2):
            raise BaseException("only unary or binary factors are supported...")
        rows = len(self.varset[0].domain)
        return (rows, self.observed_domain_size) if len(self.varset) == 1 else (rows, len(self.varset[1].domain))

    def update_message_to(self, var):
        """LBP.py:490-526.  Exact: one device op (UNARY / PAIR_TM / PAIR_MT).  Approximate
        (`use_approx_inference`): au.sparse_vec_mat_dot on the device, then renormalise."""
        e = self.graph._engine
        other_vars = [v for v in self.varset if v.id != var.id]
        dst = e.slot[str(self), str(var)]
        if len(other_vars) == 0:
            e.run_op((_ffi.OP_UNARY, int(e.topo.unary_slot[e.topo.factor_index[self.id]]), 0, dst))
            return
        o_var = other_vars[0]
        o_var_dim = self.potential_table.var_id2dim[o_var.id]
        if not self.graph.use_approx_inference:
            kind = _ffi.OP_PAIR_TM if o_var_dim == 1 else _ffi.OP_PAIR_MT
            e.run_op((kind, int(e.topo.pair_slot[e.topo.factor_index[self.id]]), e.slot[str(o_var), str(self)], dst))
            return
        # top-K form (LBP.py:506-507, 515-516): the other variable on axis 1 -> column vector against the table's columns,
        # on axis 0 -> row vector against its rows
        incoming = self.graph.messages[str(o_var), str(self)].m
        out = Message(au.sparse_vec_mat_dot(incoming if o_var_dim == 1 else incoming.T, self.potential_table.table))
        if self.graph.normalize_messages:
            out.renormalize()
        self.graph.messages[str(self), str(var)] = out

    def get_factor_beliefs(self):
        """LBP.py:528-574."""
        g = self.graph
        t0 = time.time() if g.report_times else None
        if len(self.varset) == 1:
            beliefs = au.normalize(self.potential_table.table)
            if g.report_times: g.ub_times.append(time.time() - t0)
            return beliefs
        by_axis = {}
        for v in self.varset:
            axis = self.potential_table.var_id2dim[v.id]
            if axis not in (0, 1):
                raise NotImplementedError("only supports pairwise factors..")
            by_axis[axis] = g.messages[str(v), str(self)].m
        c = np.reshape(by_axis[0], (-1, 1))               # message of the dim-0 variable as a column
        r = np.reshape(by_axis[1], (1, -1))               # message of the dim-1 variable as a row
        if g.use_approx_beliefs:
            outer, c_idx, r_idx = au.sparse_dot(c, r)
            beliefs = au.sparse_normalize(au.sparse_pointwise_multiply(outer, c_idx, r_idx, self.potential_table.table),
                                          c_idx, r_idx)
        else:
            beliefs = au.normalize(au.dense_pointwise_multiply(au.dense_dot(c, r), self.potential_table.table))
        if g.report_times: g.bb_times.append(time.time() - t0)
        return beliefs

    def get_observed_factor_as_array(self):
        """LBP.py:576-582."""
        cell = sorted([(self.potential_table.var_id2dim[v.id], v.supervised_label_index) for v in self.varset])
        return [tuple([o for d, o in cell])]

    def get_observed_factor(self):
        """LBP.py:584-589."""
        of = np.zeros_like(self.potential_table.table, dtype=DTYPE)
        of[self.get_observed_factor_as_array()[0]] = 1.0
        return of

    def get_gradient(self):
        """LBP.py:592-613: (observed - expected) cells contracted with the feature tensor, as device
        dense_dot calls: unary g^T (1,X) . phi[:, observed_dim, :] (X,F); pairwise the (1,X*X) .
        (X*X,F) form of np.tensordot."""
        g = self.cell_gradient()
        if self.graph.report_times: self.graph.sgg_times.append(0.0)
        if self.graph.report_times: gg = time.time()
        if self.potential_table.observed_dim is not None:
            phi_g = np.asarray(self.get_phi()[:, self.potential_table.observed_dim, :], dtype=DTYPE)
            grad = au.dense_dot(np.ascontiguousarray(g.T), phi_g)
        else:
            phi = np.asarray(self.get_phi(), dtype=DTYPE)
            if phi.shape[:2] != g.shape:
                raise ValueError('shape-mismatch for sum')
            grad = au.dense_dot(np.reshape(g, (1, g.size)), np.reshape(phi, (g.size, -1)))
        grad = np.reshape(grad, (1, np.size(grad)))
        if self.graph.report_times: self.graph.gg_times.append(time.time() - gg)
        return grad

    def cell_gradient(self):
        """LBP.py:615-619: onehot(observed cell) - beliefs, on the device."""
        exp_counts = np.ascontiguousarray(self.get_factor_beliefs(), dtype=DTYPE)
        cell = self.get_observed_factor_as_array()[0]
        if len(cell) == 1:
            cell = (cell[0], 0)
        flat = int(np.ravel_multi_index(cell, exp_counts.shape))
        dev = _device()
        t = torch.from_numpy(exp_counts).to(dev)
        out = torch.empty_like(t)
        _ffi.check(_ffi.lib.mlbp_observed_minus_f64(t.data_ptr(), t.numel(), flat, out.data_ptr(), _stream_ptr(dev)))
        return out.cpu().numpy()

    def cell_gradient_alt(self):
        """LBP.py:621-628 (same quantity)."""
        return self.cell_gradient()


class ObservedFactor(FactorNode):
    """LBP.py:630-634."""

    def __init__(self, id, observed_domain_type, observed_value):
        super().__init__(id, factor_type=UNARY_FACTOR, observed_domain_type=observed_domain_type, observed_value=observed_value)


class Message():
    """LBP.py:637-667: an (X,1) float64 column."""

    def __init__(self, m):
        if __debug__:
            assert isinstance(m, np.ndarray) and not (m < 0.0).any()
        self.m = m if m.shape == (m.size, 1) else m.reshape(m.size, 1)      # anything becomes a column (LBP.py:641-644)

    def __str__(self):
        return np.array_str(self.m)

    def renormalize(self):
        """LBP.py:649-659: positive total -> m / total, else uniform; one device call."""
        dev = _device()
        t = torch.from_numpy(np.ascontiguousarray(self.m, dtype=np.float64)).to(dev)
        out = torch.empty_like(t)
        _ffi.check(_ffi.lib.mlbp_normalize_f64(t.data_ptr(), out.data_ptr(), 1, t.numel(), _ffi.NORM_UNIFORM, None,
                                               _stream_ptr(dev)))
        self.m = out.cpu().numpy().reshape(np.shape(self.m))
        if __debug__: assert np.size(self.m[self.m < 0.0]) == 0
        if __debug__: assert np.abs(np.sum(self.m) - 1.0) < 1e-10

    @staticmethod
    def new_message(domain, init):
        return Message(np.full((len(domain), 1), init, dtype=DTYPE))


class PotentialTable():
    """LBP.py:670-715 (host-side indexing only)."""

    def __init__(self, v_id2dim, table=None, observed_dim=None):
        self.factor, self.observed_dim, self.var_id2dim = None, observed_dim, v_id2dim
        if table is None:
            return                                          # filled later by slice_potentials()
        if __debug__: assert isinstance(table, np.ndarray)
        if observed_dim is not None:
            if __debug__: assert len(v_id2dim) == 1
            if next(iter(v_id2dim.values())) != 0:
                raise NotImplementedError("a unary factor should always be a column vector")
        self._adopt(table)

    def _adopt(self, table):
        """Common tail of the constructor and slice_potentials (LBP.py:678-691, 702-710): an observed
        unary factor keeps one COLUMN as (X,1); tables are float64; square or column shaped."""
        if self.observed_dim is not None:
            table = np.reshape(table[:, self.observed_dim], (np.shape(table)[0], 1))
        self.table = table if table.dtype == DTYPE else table.astype(DTYPE)
        if self.table.ndim > 1:
            if __debug__: assert self.table.shape[0] == self.table.shape[1] or self.table.shape[1] == 1

    def slice_potentials(self):
        """LBP.py:695-710: pairwise tables alias the graph-level pot array (no copy)."""
        self._adopt(np.reshape(self.factor.get_pot(), self.factor.get_shape()))

    def add_factor(self, factor):
        if __debug__: assert isinstance(factor, FactorNode)
        if __debug__: assert self.factor is None
        self.factor = factor


def pointwise_multiply(m1, m2):
    """LBP.py:717-730: Message(nan_to_num(m1 * m2)), one device call."""
    if __debug__: assert isinstance(m1, Message)
    if __debug__: assert isinstance(m2, Message)
    if __debug__: assert np.shape(m1.m) == np.shape(m2.m)
    dev = _device()
    a = torch.from_numpy(np.ascontiguousarray(m1.m, dtype=np.float64)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(m2.m, dtype=np.float64)).to(dev)
    out = torch.empty_like(a)
    _ffi.check(_ffi.lib.mlbp_pointwise_multiply_f64(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), 1,
                                                    _stream_ptr(dev)))
    return Message(out.cpu().numpy().reshape(np.shape(m1.m)))


class PhiWrapper:
    """LBP.py:732-736."""

    def __init__(self, phi_en_en, phi_en_en_w1, phi_en_de):
        self.phi_en_en, self.phi_en_en_w1, self.phi_en_de = phi_en_en, phi_en_en_w1, phi_en_de


class ThetaWrapper(object):
    """LBP.py:739-745."""

    def __init__(self, theta_en_en_names, theta_en_en, theta_en_de_names, theta_en_de):
        self.theta_en_en_names, self.theta_en_en = theta_en_en_names, theta_en_en
        self.theta_en_de_names, self.theta_en_de = theta_en_de_names, theta_en_de
