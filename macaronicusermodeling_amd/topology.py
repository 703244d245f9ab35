"""Integer description of one factor-graph shape and the host-side (C++) logic over it.

`GraphTopology` is what a batch of graphs shares: which variables each factor joins, on which table
axis, and each variable's factor list in creation order.  It marshals that into the flat arrays of
`mlbp_topology` (include/mlbp.h) and exposes the bit-exact host functions of libmlbp.so:
loop test (LBP.py:174-190), BFS message schedule (LBP.py:155-172), message-slot numbering and the
per-root sweep compiler (LBP.py:223-243).  No float work and no GPU needed here.
"""
import ctypes as C

import numpy as np

from . import _ffi


class GraphTopology:
    def __init__(self, factors, var_ids=None, facsets=None):
        """factors: iterable of (factor_id, [var ids in varset order], [table axis per var]) in
        CREATION order (the order `add_varset_with_potentials` was called: it fixes each
        variable's facset order, LBP.py:449-452).  var_ids: optional explicit variable order (the
        insertion order of FactorGraph.variables); defaults to first appearance.  facsets: optional
        {var id: [factor ids]} giving each variable's neighbour order explicitly (the object API
        passes VariableNode.facset); defaults to creation order."""
        factors = [(int(fid), [int(v) for v in vs], [int(d) for d in ds]) for fid, vs, ds in factors]
        if not factors:
            raise ValueError('a graph needs at least one factor (LBP.py:194)')
        ids = [f[0] for f in factors]
        if len(set(ids)) != len(ids):
            raise ValueError('factor ids must be unique')
        if var_ids is None:
            var_ids = []
            for _, vs, _ in factors:
                for v in vs:
                    if v not in var_ids:
                        var_ids.append(v)
        self.var_ids = [int(v) for v in var_ids]
        self.var_index = {v: i for i, v in enumerate(self.var_ids)}
        by_id = sorted(factors, key=lambda f: f[0])                  # LBP.py:195-196
        self.factor_ids = [f[0] for f in by_id]
        self.factor_index = {fid: i for i, fid in enumerate(self.factor_ids)}
        F, n = len(by_id), len(self.var_ids)
        self.n_vars, self.n_factors = n, F
        self.fac_nvars = np.array([len(f[1]) for f in by_id], dtype=np.int32)
        self.fac_var = np.full(2 * F, -1, dtype=np.int32)
        self.fac_dim = np.full(2 * F, -1, dtype=np.int32)
        for j, (_, vs, ds) in enumerate(by_id):
            if len(vs) > 2:
                raise NotImplementedError('Currently supporting unary and pairwise factors...')  # LBP.py:447
            for k, (v, d) in enumerate(zip(vs, ds)):
                self.fac_var[2 * j + k] = self.var_index[v]
                self.fac_dim[2 * j + k] = d
        if facsets is None:
            fs = [[] for _ in range(n)]
            for fid, vs, _ in factors:                               # creation order
                for v in vs:
                    fs[self.var_index[v]].append(self.factor_index[fid])
        else:
            fs = [[self.factor_index[int(fid)] for fid in facsets[v]] for v in self.var_ids]
        facsets = fs
        self.facsets = facsets
        self.var_fac_off = np.zeros(n + 1, dtype=np.int32)
        self.var_fac_off[1:] = np.cumsum([len(s) for s in facsets])
        self.var_fac = np.array([f for s in facsets for f in s] or [0], dtype=np.int32)
        self._c = _ffi.Topology(n, F, _ffi.i32ptr(self.fac_nvars), _ffi.i32ptr(self.fac_var),
                                _ffi.i32ptr(self.fac_dim), _ffi.i32ptr(self.var_fac_off),
                                _ffi.i32ptr(self.var_fac))
        # message slots
        self.f2v = np.empty(2 * F, dtype=np.int32)
        self.v2f = np.empty(2 * F, dtype=np.int32)
        self.pair_slot = np.empty(F, dtype=np.int32)
        self.unary_slot = np.empty(F, dtype=np.int32)
        self.n_msgs = _ffi.check(_ffi.lib.mlbp_message_slots(
            C.byref(self._c), _ffi.i32ptr(self.f2v), _ffi.i32ptr(self.v2f), _ffi.i32ptr(self.pair_slot),
            _ffi.i32ptr(self.unary_slot)))
        self.P = int((self.fac_nvars == 2).sum())
        self.U = int((self.fac_nvars == 1).sum())
        self.pair_factors = [j for j in range(F) if self.fac_nvars[j] == 2]    # factor index by pair slot
        self.unary_factors = [j for j in range(F) if self.fac_nvars[j] == 1]
        # incoming factor->variable slots per variable, facset order (marginals, LBP.py:394-396)
        in_slots, off = [], [0]
        for v in range(n):
            for j in facsets[v]:
                k = 0 if self.fac_var[2 * j] == v else 1
                in_slots.append(int(self.f2v[2 * j + k]))
            off.append(len(in_slots))
        self.in_off = np.array(off, dtype=np.int32)
        self.in_slots = np.array(in_slots or [0], dtype=np.int32)
        self._sweep_cache = {}

    # ---- naming, as the reference prints nodes (LBP.py:357-358, 430-431) ----
    def node_name(self, node):
        return 'X_%d' % self.var_ids[node] if node < self.n_vars else 'F_%d' % self.factor_ids[node - self.n_vars]

    def slot_keys(self):
        """(src name, dst name) of every message slot, in slot order."""
        keys = [None] * self.n_msgs
        for j in range(self.n_factors):
            fn = 'F_%d' % self.factor_ids[j]
            for k in range(int(self.fac_nvars[j])):
                vn = 'X_%d' % self.var_ids[self.fac_var[2 * j + k]]
                keys[self.f2v[2 * j + k]] = (fn, vn)
                if self.v2f[2 * j + k] >= 0:
                    keys[self.v2f[2 * j + k]] = (vn, fn)
        return keys

    # ---- host logic ----
    def has_loops(self, root_var_id):
        return bool(_ffi.check(_ffi.lib.mlbp_has_loops(C.byref(self._c), self.var_index[root_var_id])))

    def message_schedule(self, root_var_id):
        """[(child node, parent node)] with nodes as dense codes (variable v -> v, factor f ->
        n_vars + f); use node_name() for the reference's 'X_i' / 'F_j' strings."""
        cap = 4 * (self.n_factors + self.n_vars) + 8
        buf = np.empty(2 * cap, dtype=np.int32)
        n = _ffi.check(_ffi.lib.mlbp_message_schedule(C.byref(self._c), self.var_index[root_var_id],
                                                      _ffi.i32ptr(buf), cap))
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

    def compile_sweep(self, root_var_id):
        """(ops [n,4] int32, srcs int32) of one sweep rooted at the variable; cached per root."""
        key = int(root_var_id)
        if key not in self._sweep_cache:
            cap_ops = 8 * (self.n_factors + self.n_vars) + 8
            cap_srcs = cap_ops * max(1, max(len(s) for s in self.facsets))
            ops = np.empty(4 * cap_ops, dtype=np.int32)
            srcs = np.empty(cap_srcs, dtype=np.int32)
            ns = C.c_int32(0)
            n = _ffi.check(_ffi.lib.mlbp_compile_sweep(C.byref(self._c), self.var_index[key], _ffi.i32ptr(ops),
                                                       cap_ops, _ffi.i32ptr(srcs), cap_srcs, C.byref(ns)))
            self._sweep_cache[key] = (ops[:4 * n].reshape(n, 4).copy(), srcs[:ns.value].copy())
        return self._sweep_cache[key]

    def compile_program(self, roots):
        """Concatenated op list for a root sequence; equal roots share their ops.
        Returns (ops [n,4], srcs, sweeps [S,2])."""
        ops_all, srcs_all, sweeps, where = [], [], [], {}
        n_ops = n_srcs = 0
        for r in roots:
            r = int(r)
            if r not in where:
                ops, srcs = self.compile_sweep(r)
                ops = ops.copy()
                is_var = ops[:, 0] == _ffi.OP_VAR
                ops[is_var, 1] += n_srcs
                where[r] = (n_ops, len(ops))
                ops_all.append(ops)
                srcs_all.append(srcs)
                n_ops += len(ops)
                n_srcs += len(srcs)
            sweeps.append(where[r])
        ops = np.concatenate(ops_all).astype(np.int32)
        srcs = np.concatenate(srcs_all).astype(np.int32) if n_srcs else np.zeros(0, dtype=np.int32)
        return ops, srcs, np.array(sweeps, dtype=np.int32).reshape(-1, 2)

    def plan(self, roots):
        """What the host-side program rewrites make of this root sequence (no GPU needed): dict with the update
        counts of the fused form and the tile budget of the shared-table form (mlbp_program_plan)."""
        ops, srcs, sweeps = self.compile_program(roots)
        out = np.zeros(8, dtype=np.int32)
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        srcs_p = np.ascontiguousarray(srcs if len(srcs) else np.zeros(1), dtype=np.int32)
        sweeps = np.ascontiguousarray(sweeps, dtype=np.int32)
        _ffi.check(_ffi.lib.mlbp_program_plan(_ffi.i32ptr(ops), len(ops), _ffi.i32ptr(srcs_p), len(srcs), _ffi.i32ptr(sweeps),
                                              len(sweeps), self.n_msgs, self.P, self.U, _ffi.i32ptr(out)))
        keys = ('updates', 'lone_variable_updates', 'fused_updates', 'bundles', 'shared_ok', 'shared_tiles',
                'shared_updates', 'shared_tile_bytes')
        plan = dict(zip(keys, (int(v) for v in out)))
        plan['shared_product_fused'] = (plan['shared_ok'] >> 1) & 1       # the message's producer stores c (.) message: K2, K3, chains, rings
        plan['shared_gradient_from_tiles'] = (plan['shared_ok'] >> 2) & 1
        plan['shared_product_fused3'] = (plan['shared_ok'] >> 3) & 1      # ... stores sqrt(c) (.) message: variables with three pairwise factors (K4)
        plan['shared_ok'] &= 1
        return plan

    @classmethod
    def from_spec(cls, spec):
        """From the plain-data specs used by the tests and the benchmark (tests/golden/cases.py)."""
        return cls([(f['id'], f['vars'], f['dims']) for f in spec['factors']])
