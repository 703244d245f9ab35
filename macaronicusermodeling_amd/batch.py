"""FactorGraphBatch: B independent factor graphs of one shape, resident in HBM, swept by libmlbp.so.

This is the performance form of `FactorGraph.initialize` + `treelike_inference` +
`get_marginal` / `get_posterior_probs` (LBP.py:192-259): the same algorithm and update order,
batched over graphs.  torch is used only to own device memory and the HIP stream; every number is
produced by the HIP kernels behind the C ABI (include/mlbp.h).

HBM layout (all float64, the reference dtype):
    pair_tables  [n_pair_tables][X][X]   row-major; a graph's pairwise factor p reads table
                                         pair_tab[b][p]  (unique per (graph, factor), or shared)
    unary_tables [n_unary_tables][X]     a unary factor's (X,1) table as one contiguous row
    msgs         [B][n_msgs][X]          slot order = GraphTopology.slot_keys()
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .topology import GraphTopology


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Program:
    """Validated device-resident op list for one root sequence (mlbp_program_create)."""

    def __init__(self, topo, roots):
        self.roots = tuple(int(r) for r in roots)
        ops, srcs, sweeps = topo.compile_program(self.roots)
        self.n_ops = len(ops)
        self.ops, self.srcs, self.sweeps = ops, srcs, sweeps
        h = C.c_void_p()
        ops_c = np.ascontiguousarray(ops.reshape(-1))
        srcs_c = np.ascontiguousarray(srcs if len(srcs) else np.zeros(1, dtype=np.int32))
        sw_c = np.ascontiguousarray(sweeps.reshape(-1))
        _ffi.check(_ffi.lib.mlbp_program_create(_ffi.i32ptr(ops_c), len(ops), _ffi.i32ptr(srcs_c), len(srcs),
                                                _ffi.i32ptr(sw_c), len(sweeps), topo.n_msgs, topo.P, topo.U,
                                                C.byref(h)))
        self.handle = h

    def status(self):
        return _ffi.check(_ffi.lib.mlbp_program_status(self.handle))

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h is not None and h.value:
            _ffi.lib.mlbp_program_destroy(h)
            self.handle = None


class FactorGraphBatch:
    def __init__(self, topo, X, B, device='cuda:0', normalize_messages=True):
        if not isinstance(topo, GraphTopology):
            raise TypeError('topo must be a GraphTopology')
        self.topo, self.X, self.B = topo, int(X), int(B)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('FactorGraphBatch lives on an MI355X; there is no CPU path')
        self.normalize_messages = bool(normalize_messages)
        self.msgs = torch.empty(self.B, topo.n_msgs, self.X, dtype=torch.float64, device=self.device)
        self.pair_tables = self.pair_tab = self.unary_tables = self.unary_tab = None
        self._in_off = torch.from_numpy(topo.in_off).to(self.device)
        self._in_slots = torch.from_numpy(topo.in_slots).to(self.device)
        self._programs = {}
        self.is_loopy = None

    # ---- tables -------------------------------------------------------------------------------
    @staticmethod
    def _as_index(idx, B, n, limit, what, device):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64).reshape(B, n))
        if idx.size and (idx.min() < 0 or idx.max() >= limit):
            raise IndexError('%s index out of range [0, %d)' % (what, limit))
        return torch.from_numpy(idx.astype(np.int32)).to(device)

    def set_pair_tables(self, tables, pair_tab=None):
        """tables: [n][X][X] float64 (numpy or torch; copied to the device if needed).
        pair_tab: [B][P] integer table index per graph and pair slot; default = unique tables
        b*P + p."""
        t = torch.as_tensor(tables, dtype=torch.float64).to(self.device).contiguous()
        if t.dim() != 3 or t.shape[1] != self.X or t.shape[2] != self.X:
            raise ValueError('pair tables must be [n][X][X]')
        if pair_tab is None:
            if t.shape[0] != self.B * self.topo.P:
                raise ValueError('need B*P tables when pair_tab is omitted')
            pair_tab = np.arange(self.B * self.topo.P).reshape(self.B, self.topo.P)
        self.pair_tab = self._as_index(pair_tab, self.B, self.topo.P, t.shape[0], 'pair table', self.device)
        self.pair_tables = t

    def set_unary_tables(self, tables, unary_tab=None):
        t = torch.as_tensor(tables, dtype=torch.float64).to(self.device).contiguous()
        if t.dim() != 2 or t.shape[1] != self.X:
            raise ValueError('unary tables must be [n][X]')
        if unary_tab is None:
            if t.shape[0] != self.B * self.topo.U:
                raise ValueError('need B*U tables when unary_tab is omitted')
            unary_tab = np.arange(self.B * self.topo.U).reshape(self.B, self.topo.U)
        self.unary_tab = self._as_index(unary_tab, self.B, self.topo.U, t.shape[0], 'unary table', self.device)
        self.unary_tables = t

    # ---- FactorGraph.initialize (LBP.py:192-216) ----------------------------------------------------
    def initialize(self, loop_root=None):
        root = self.topo.var_ids[0] if loop_root is None else loop_root
        self.is_loopy = self.topo.has_loops(root)
        _ffi.check(_ffi.lib.mlbp_init_messages_f64(self.msgs.data_ptr(), self.B * self.topo.n_msgs, self.X,
                                                   _stream_ptr(self.device)))

    # ---- FactorGraph.treelike_inference (LBP.py:218-245) ------------------------------------------
    def program(self, roots):
        key = tuple(int(r) for r in roots)
        if key not in self._programs:
            self._programs[key] = Program(self.topo, key)
        return self._programs[key]

    def sweep(self, roots, init=False):
        """Runs len(roots) sweeps, sweep s rooted at variable id roots[s], on every graph, in one
        launch.  init=True starts from uniform messages (initialize() fused into the launch)."""
        prog = self.program(roots)
        a = _ffi.SweepArgs()
        a.B, a.X = self.B, self.X
        if self.topo.P:
            if self.pair_tables is None:
                raise RuntimeError('set_pair_tables() first')
            a.n_pair_tables = self.pair_tables.shape[0]
            a.pair_tables, a.pair_tab = self.pair_tables.data_ptr(), self.pair_tab.data_ptr()
        if self.topo.U:
            if self.unary_tables is None:
                raise RuntimeError('set_unary_tables() first')
            a.n_unary_tables = self.unary_tables.shape[0]
            a.unary_tables, a.unary_tab = self.unary_tables.data_ptr(), self.unary_tab.data_ptr()
        a.msgs = self.msgs.data_ptr()
        a.normalize_messages = 1 if self.normalize_messages else 0
        a.init_messages = 1 if init else 0
        _ffi.check(_ffi.lib.mlbp_sweep_f64(prog.handle, C.byref(a), _stream_ptr(self.device)))
        return prog

    def treelike_inference(self, iterations, roots):
        """`iterations` sweeps if the graph is loopy, else one (LBP.py:219); `roots` replaces the
        per-sweep random.sample draw (LBP.py:223).  Returns the number of sweeps run."""
        if self.is_loopy is None:
            raise RuntimeError('initialize() first')
        n = iterations if self.is_loopy else 1
        if len(roots) < n:
            raise ValueError('need %d roots, got %d' % (n, len(roots)))
        self.sweep(list(roots)[:n])
        return n

    # ---- read-outs --------------------------------------------------------------------------------
    def marginals(self, out=None):
        """[B][n_vars][X] in GraphTopology.var_ids order (VariableNode.get_marginal, LBP.py:392-400)."""
        if out is None:
            out = torch.empty(self.B, self.topo.n_vars, self.X, dtype=torch.float64, device=self.device)
        _ffi.check(_ffi.lib.mlbp_marginals_f64(self.msgs.data_ptr(), self.B, self.topo.n_msgs, self.X,
                                               self.topo.n_vars, self._in_off.data_ptr(),
                                               self._in_slots.data_ptr(), 1 if self.normalize_messages else 0,
                                               out.data_ptr(), _stream_ptr(self.device)))
        return out

    def log_posterior(self, labels, marginals=None):
        """[B] sum over variables of log marginal[label] (FactorGraph.get_posterior_probs,
        LBP.py:247-259).  labels: [B][n_vars] integer label indices."""
        lab = np.asarray(labels, dtype=np.int64).reshape(self.B, self.topo.n_vars)
        if lab.min() < 0 or lab.max() >= self.X:
            raise IndexError('label index out of range')
        lab_d = torch.from_numpy(lab.astype(np.int32)).to(self.device)
        m = self.marginals() if marginals is None else marginals
        out = torch.empty(self.B, dtype=torch.float64, device=self.device)
        _ffi.check(_ffi.lib.mlbp_log_posterior_f64(m.data_ptr(), lab_d.data_ptr(), self.B, self.topo.n_vars,
                                                   self.X, out.data_ptr(), _stream_ptr(self.device)))
        return out
