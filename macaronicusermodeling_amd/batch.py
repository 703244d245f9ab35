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
import threading

import numpy as np
import torch

from . import _ffi
from .topology import GraphTopology


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Program:
    """Validated device-resident op list for one root sequence (mlbp_program_create)."""

    def __init__(self, topo, roots, max_graphs=0):
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
        _ffi.check(_ffi.lib.mlbp_program_set_readout(h, topo.n_vars, _ffi.i32ptr(topo.in_off), _ffi.i32ptr(topo.in_slots)))
        if max_graphs > 0:
            _ffi.check(_ffi.lib.mlbp_program_reserve(h, int(max_graphs)))

    def status(self):
        return _ffi.check(_ffi.lib.mlbp_program_status(self.handle))

    def exact_count(self, B):
        """Graphs of the last launch that needed the exact kernel (mlbp_program_exact_count)."""
        return _ffi.check(_ffi.lib.mlbp_program_exact_count(self.handle, B))

    def skippable_updates(self):
        """Updates of the root sequence that skip_unchanged drops (mlbp_program_skippable_updates)."""
        return _ffi.check(_ffi.lib.mlbp_program_skippable_updates(self.handle))

    def __del__(self):
        h = getattr(self, 'handle', None)
        lib = getattr(_ffi, 'lib', None) if _ffi is not None else None      # module may be gone at interpreter exit
        if h is not None and h.value and lib is not None:
            lib.mlbp_program_destroy(h)
            self.handle = None


# Programs shared between the one-graph batches of the object API (LBP.py drop-in): every TrainingInstance of one sentence
# shape builds the same topology and draws its roots from the same few variables, and creating a Program costs a dozen
# synchronous allocations and copies.  Per thread (a program's scratch buffers belong to one stream at a time,
# include/mlbp.h), keyed by the topology's integer description, the device and the root sequence; bounded.
_shared_programs = threading.local()
_SHARED_PROGRAMS_MAX = 512


def _topology_signature(topo):
    return (topo.n_vars, tuple(topo.var_ids), topo.fac_nvars.tobytes(), topo.fac_var.tobytes(), topo.fac_dim.tobytes(),
            tuple(tuple(f) for f in topo.facsets))



class FactorGraphBatch:
    def __init__(self, topo, X, B, device='cuda:0', normalize_messages=True, use_approx_inference=False, use_approx_beliefs=False,
                 share_programs=False):
        if not isinstance(topo, GraphTopology):
            raise TypeError('topo must be a GraphTopology')
        self.topo, self.X, self.B = topo, int(X), int(B)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('FactorGraphBatch lives on an MI355X; there is no CPU path')
        self.normalize_messages = bool(normalize_messages)
        # FactorGraph.use_approx_inference / use_approx_beliefs (LBP.py:52-53): top-100 variants, selected on the device
        self.use_approx_inference, self.use_approx_beliefs = bool(use_approx_inference), bool(use_approx_beliefs)
        self.skip_unchanged = False       # MLBP_SWEEP_SKIP_UNCHANGED on every sweep (same output bits, fewer updates)
        self.msgs = torch.empty(self.B, topo.n_msgs, self.X, dtype=torch.float64, device=self.device)
        self.pair_tables = self.pair_tab = self.unary_tables = self.unary_tab = None
        self._in_off = torch.from_numpy(topo.in_off).to(self.device)
        self._in_slots = torch.from_numpy(topo.in_slots).to(self.device)
        self._programs = {}
        self._share_programs = bool(share_programs)     # the object API: Programs come from the per-thread shared cache
        self.is_loopy = None

    # ---- tables -------------------------------------------------------------------------------
    @staticmethod
    def _as_index(idx, B, n, limit, what, device):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64).reshape(B, n))
        if idx.size and (idx.min() < 0 or idx.max() >= limit):
            raise IndexError('%s index out of range [0, %d)' % (what, limit))
        return torch.from_numpy(idx.astype(np.int32)).to(device)

    def set_pair_tables(self, tables, pair_tab=None, dtype=torch.float64):
        """tables: [n][X][X] float64 (numpy or torch; copied to the device if needed).
        pair_tab: [B][P] integer table index per graph and pair slot; default = unique tables
        b*P + p.  dtype=torch.float32 keeps the tables in float32 on the device (X = 256 or 512 only:
        the large-state mode, half the bytes per update; sweeps and marginals only)."""
        if dtype not in (torch.float64, torch.float32):
            raise TypeError('pair tables are float64 or float32')
        if dtype == torch.float32 and self.X not in (256, 512):
            raise ValueError('float32 pair tables need X = 256 or 512')
        t = torch.as_tensor(tables).to(self.device).to(dtype).contiguous()
        if t.dim() != 3 or t.shape[1] != self.X or t.shape[2] != self.X:
            raise ValueError('pair tables must be [n][X][X]')
        self._pair_dense = pair_tab is None          # identity index: stated to the kernel (MLBP_SWEEP_DENSE_TABLES)
        if pair_tab is None:
            if t.shape[0] != self.B * self.topo.P:
                raise ValueError('need B*P tables when pair_tab is omitted')
            pair_tab = np.arange(self.B * self.topo.P).reshape(self.B, self.topo.P)
        host_tab = np.asarray(pair_tab).reshape(self.B, self.topo.P)
        # every graph reads the same table for factor p (the reference's layout: one pot array behind all
        # pairwise factors): eligible for the shared-table kernel, which re-checks it on the device
        self.pair_tables_shared = bool(self.topo.P) and bool((host_tab == host_tab[:1]).all())
        self._pair_row_host = np.ascontiguousarray(host_tab[0], dtype=np.int32) if self.pair_tables_shared else None
        self.pair_tab = self._as_index(host_tab, self.B, self.topo.P, t.shape[0], 'pair table', self.device)
        self.pair_tables = t

    def set_unary_tables(self, tables, unary_tab=None):
        t = torch.as_tensor(tables, dtype=torch.float64).to(self.device).contiguous()
        if t.dim() != 2 or t.shape[1] != self.X:
            raise ValueError('unary tables must be [n][X]')
        self._unary_dense = unary_tab is None
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
            if self._share_programs:
                cache = getattr(_shared_programs, 'by_key', None)
                if cache is None:
                    cache = _shared_programs.by_key = {}
                if not hasattr(self, '_topo_sig'):
                    self._topo_sig = (_topology_signature(self.topo), self.device.index)
                full = (self._topo_sig, key)
                prog = cache.get(full)
                if prog is None:
                    if len(cache) >= _SHARED_PROGRAMS_MAX:
                        cache.clear()             # programs still referenced by a batch's own dict live on
                    prog = cache[full] = Program(self.topo, key, max_graphs=self.B)
                self._programs[key] = prog
            else:
                self._programs[key] = Program(self.topo, key, max_graphs=self.B)
        return self._programs[key]

    def sweep(self, roots, init=False, marginals=None, gradient=None, keep_messages=True, skip_unchanged=None, posterior=None, _collect=None):
        """Runs len(roots) sweeps, sweep s rooted at variable id roots[s], on every graph, in one
        launch.  init=True starts from uniform messages (initialize() fused into the launch);
        marginals: optional [B][n_vars][X] device tensor that receives every variable's marginal
        after the last sweep, read out of the on-chip messages in the same launch; gradient: optional
        (out_ee [B][F_ee], out_ed [B][F_ed]) device tensors that receive the per-graph gradients
        (set_features / set_observations first), fused into the launch when the kernel allows;
        keep_messages=False lets a launch whose read-outs (marginals, gradient) are fused skip the write-back of
        self.msgs (their contents are then undefined); posterior: optional (labels int32 [B][n_vars], out [B], sum_out [1] or
        None) device tensors -- get_posterior_probs of every graph (LBP.py:247-259) and their batch sum behind the sweeps of
        the same call (needs `marginals`; on the fast X = 64 paths the fix-up launch takes it);
        skip_unchanged (default: self.skip_unchanged, False) drops the
        updates of the root sequence that would recompute a message from unchanged inputs (MLBP_SWEEP_SKIP_UNCHANGED:
        same output bits, fewer updates)."""
        prog = self.program(roots)
        a = _ffi.SweepArgs()
        a.B, a.X = self.B, self.X
        if self.topo.P:
            if self.pair_tables is None:
                raise RuntimeError('set_pair_tables() first')
            a.n_pair_tables = self.pair_tables.shape[0]
            a.pair_tab = self.pair_tab.data_ptr()
            if self.pair_tables.dtype == torch.float32:
                a.pair_tables_f32 = self.pair_tables.data_ptr()
                a.flags |= _ffi.SWEEP_PAIR_TABLES_F32
            else:
                a.pair_tables = self.pair_tables.data_ptr()
        if self.topo.U:
            if self.unary_tables is None:
                raise RuntimeError('set_unary_tables() first')
            a.n_unary_tables = self.unary_tables.shape[0]
            a.unary_tables, a.unary_tab = self.unary_tables.data_ptr(), self.unary_tab.data_ptr()
        a.msgs = self.msgs.data_ptr()
        a.normalize_messages = 1 if self.normalize_messages else 0
        a.init_messages = 1 if init else 0
        if self.use_approx_inference:
            a.flags |= _ffi.SWEEP_APPROX_INFERENCE
        if getattr(self, 'skip_unchanged', False) if skip_unchanged is None else skip_unchanged:
            a.flags |= _ffi.SWEEP_SKIP_UNCHANGED
        if (getattr(self, '_pair_dense', False) or not self.topo.P) and (getattr(self, '_unary_dense', False) or not self.topo.U):
            a.flags |= _ffi.SWEEP_DENSE_TABLES
        if getattr(self, 'pair_tables_shared', False):
            a.flags |= _ffi.SWEEP_SHARED_PAIR_TABLES
            if getattr(self, '_pair_row_host', None) is not None:       # X >= 128: the update-by-update contraction path needs the row on the host
                a.pair_tab_host = self._pair_row_host.ctypes.data
        if not keep_messages and (marginals is not None or gradient is not None):
            a.flags |= _ffi.SWEEP_NO_MESSAGE_WRITEBACK      # honoured by the kernels whose read-outs are fused (lean, shared)
        if marginals is not None:
            if tuple(marginals.shape) != (self.B, self.topo.n_vars, self.X) or marginals.dtype != torch.float64:
                raise ValueError('marginals must be float64 [B][n_vars][X]')
            a.marginals = marginals.data_ptr()
        if gradient is not None:
            ga = self._gradient_args(*gradient)
            a.gradient = C.addressof(ga)
        if posterior is not None:
            lab, out, tot = posterior
            if marginals is None or lab.dtype != torch.int32 or tuple(lab.shape) != (self.B, self.topo.n_vars) or out.numel() < self.B:
                raise ValueError('posterior needs marginals, int32 labels [B][n_vars] and an output of B doubles')
            pa = _ffi.PosteriorArgs()
            pa.labels, pa.out, pa.sum_out = lab.data_ptr(), out.data_ptr(), None if tot is None else tot.data_ptr()
            a.posterior = C.addressof(pa)
        if _collect is not None:                  # sweep_groups(): gather instead of launching
            _collect.append((prog, a, gradient and ga))
            return prog
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

    # ---- beliefs / gradient (LBP.py:528-619, 301-320) ---------------------------------------------
    def _need_f64_tables(self, what):
        if self.pair_tables is not None and self.pair_tables.dtype != torch.float64:
            raise NotImplementedError('%s needs float64 pairwise tables' % what)

    def _pair_slots(self):
        """device int32 [P] message slots of the dim-0 / dim-1 variable -> factor messages."""
        if not hasattr(self, '_c_slot'):
            topo = self.topo
            c, r = [], []
            for j in topo.pair_factors:
                k0 = 0 if topo.fac_dim[2 * j] == 0 else 1          # varset position sitting on table axis 0
                c.append(int(topo.v2f[2 * j + k0]))
                r.append(int(topo.v2f[2 * j + 1 - k0]))
            self._c_slot = torch.tensor(c or [0], dtype=torch.int32, device=self.device)
            self._r_slot = torch.tensor(r or [0], dtype=torch.int32, device=self.device)
        return self._c_slot, self._r_slot

    def pair_beliefs(self):
        self._need_f64_tables('pair_beliefs')
        """[B][P][X][X]: FactorNode.get_factor_beliefs of every pairwise factor (LBP.py:543-569)."""
        c, r = self._pair_slots()
        out = torch.empty(self.B, self.topo.P, self.X, self.X, dtype=torch.float64, device=self.device)
        _ffi.check(_ffi.lib.mlbp_pair_beliefs_f64(self.msgs.data_ptr(), self.B, self.topo.n_msgs, self.X, self.topo.P,
                                                  self.pair_tables.data_ptr(), self.pair_tab.data_ptr(),
                                                  self.pair_tables.shape[0], c.data_ptr(), r.data_ptr(),
                                                  out.data_ptr(), _stream_ptr(self.device)))
        return out

    def set_features(self, phi_en_en, phi_en_en_w1, phi_en_de, pair_phi, unary_kind, share_with=None):
        """Feature tensors (X,X,F_ee) x2 and (X,Vde,F_ed) shared by the batch, and the per-slot
        selector FactorNode.get_phi implies (LBP.py:469-480): pair_phi[p] in {0: gap > 1, 1: gap == 1},
        unary_kind[u] in {0, 1 (en_en by gap), 2 (en_de)}."""
        dev = self.device
        if share_with is not None:
            # the other batch's device copies (and their two re-layouts), read-only: the buckets of a TiDirTrainer -- one batch per
            # sentence shape, hundreds of them -- read ONE set of feature tensors, and a grouped launch writes the gradient's
            # weighted table fragments once per distinct set (mlbp_sweep_groups_f64), not once per bucket
            if share_with.device != dev:
                raise ValueError('feature tensors can only be shared by batches of one device')
            self.phi_en_en, self.phi_en_en_w1, self.phi_en_de = share_with.phi_en_en, share_with.phi_en_en_w1, share_with.phi_en_de
            self._phi_t, self._phi_p = share_with._phi_t, share_with._phi_p
        else:
            self.phi_en_en = torch.as_tensor(phi_en_en, dtype=torch.float64).to(dev).contiguous()
            self.phi_en_en_w1 = torch.as_tensor(phi_en_en_w1, dtype=torch.float64).to(dev).contiguous()
            self.phi_en_de = torch.as_tensor(phi_en_de, dtype=torch.float64).to(dev).contiguous()
            # transposed copies [column][x][F] (a layout change only): contiguous feature slabs for unary factors
            self._phi_t = tuple(p.permute(1, 0, 2).contiguous() for p in (self.phi_en_en, self.phi_en_en_w1, self.phi_en_de))
            self._phi_p = tuple(p.permute(2, 0, 1).contiguous() for p in (self.phi_en_en, self.phi_en_en_w1))   # [F][X][X]
        X = self.X
        if tuple(self.phi_en_en.shape[:2]) != (X, X) or self.phi_en_en_w1.shape != self.phi_en_en.shape or \
                self.phi_en_de.shape[0] != X:
            raise ValueError('feature tensors must be (X,X,F_ee), (X,X,F_ee), (X,Vde,F_ed)')
        pp = np.asarray(pair_phi, dtype=np.int64).reshape(self.topo.P)
        uk = np.asarray(unary_kind, dtype=np.int64).reshape(self.topo.U)
        if (pp.size and (pp.min() < 0 or pp.max() > 1)) or (uk.size and (uk.min() < 0 or uk.max() > 2)):
            raise ValueError('pair_phi must be 0/1 and unary_kind 0/1/2')
        self._pair_phi = torch.from_numpy(np.ascontiguousarray(pp if pp.size else np.zeros(1)).astype(np.int32)).to(dev)
        self._unary_kind = torch.from_numpy(np.ascontiguousarray(uk if uk.size else np.zeros(1)).astype(np.int32)).to(dev)
        self._unary_kind_host = uk

    def set_observations(self, var_labels, unary_obs):
        """var_labels [B][n_vars]: supervised label index per variable (GraphTopology.var_ids order);
        unary_obs [B][U]: observed column of each unary factor (PotentialTable.observed_dim)."""
        topo, B, X = self.topo, self.B, self.X
        lab = np.asarray(var_labels, dtype=np.int64).reshape(B, topo.n_vars)
        obs = np.asarray(unary_obs, dtype=np.int64).reshape(B, topo.U)
        if lab.min() < 0 or lab.max() >= X:
            raise IndexError('label index out of range')
        Vde = int(self.phi_en_de.shape[1])
        for u in range(topo.U):
            lim = Vde if self._unary_kind_host[u] == 2 else X
            if obs[:, u].min() < 0 or obs[:, u].max() >= lim:
                raise IndexError('observed column out of range for unary slot %d' % u)
        pl = np.zeros((B, max(topo.P, 1), 2), dtype=np.int32)
        for p, j in enumerate(topo.pair_factors):
            k0 = 0 if topo.fac_dim[2 * j] == 0 else 1
            pl[:, p, 0] = lab[:, topo.fac_var[2 * j + k0]]
            pl[:, p, 1] = lab[:, topo.fac_var[2 * j + 1 - k0]]
        ul = np.zeros((B, max(topo.U, 1)), dtype=np.int32)
        for u, j in enumerate(topo.unary_factors):
            ul[:, u] = lab[:, topo.fac_var[2 * j]]
        dev = self.device
        self._labels = torch.from_numpy(lab.astype(np.int32)).to(dev)
        self._pair_label = torch.from_numpy(pl).to(dev)
        self._unary_label = torch.from_numpy(ul).to(dev)
        self._unary_obs = torch.from_numpy(np.ascontiguousarray(obs if topo.U else np.zeros((B, 1))).astype(np.int32)).to(dev)

    def set_unary_rows(self, row_kind, row_obs):
        """States, per row of the unary table array, the phi selector (0 / 1 / 2) and observed column of every
        factor that reads the row.  With shared pairwise tables the gradient then computes each row's expected
        features once (mlbp_unary_expectations_f64) and gathers them per factor."""
        n = int(self.unary_tables.shape[0])
        rk = np.asarray(row_kind, dtype=np.int32).reshape(-1)
        ro = np.asarray(row_obs, dtype=np.int32).reshape(-1)
        if rk.shape[0] != n or ro.shape[0] != n:
            raise ValueError('need one (kind, column) per unary table row')
        self._row_kind = torch.from_numpy(rk).to(self.device)
        self._row_obs = torch.from_numpy(ro).to(self.device)
        self._uexp = torch.zeros(n, 8, dtype=torch.float64, device=self.device)

    def _derive_unary_rows(self):
        """set_unary_rows() from (unary_tab, unary_kind, unary_obs) when every row has ONE (kind, column)."""
        if getattr(self, '_row_kind', None) is not None or not self.topo.U or self.X != 64:
            return
        tab = self.unary_tab.cpu().numpy().astype(np.int64)
        obs = self._unary_obs.cpu().numpy().astype(np.int64)
        kind = np.tile(np.asarray(self._unary_kind_host, dtype=np.int64), (self.B, 1))
        n = int(self.unary_tables.shape[0])
        rk, ro = np.full(n, -1, dtype=np.int64), np.full(n, -1, dtype=np.int64)
        rk[tab.ravel()], ro[tab.ravel()] = kind.ravel(), obs.ravel()
        if not ((rk[tab] == kind).all() and (ro[tab] == obs).all()):
            self._row_kind = False                      # a row is read with two different columns: no shortcut
            return
        rk[rk < 0], ro[ro < 0] = 0, 0
        self.set_unary_rows(rk, ro)

    def _gradient_args(self, out_ee, out_ed):
        self._need_f64_tables('the gradient')
        topo = self.topo
        F_ee, F_ed = int(self.phi_en_en.shape[2]), int(self.phi_en_de.shape[2])
        if tuple(out_ee.shape) != (self.B, F_ee) or tuple(out_ed.shape) != (self.B, F_ed):
            raise ValueError('gradient outputs must be [B][F_ee] and [B][F_ed]')
        c, r = self._pair_slots()
        a = _ffi.GradientArgs()
        a.B, a.X, a.n_msgs, a.P, a.U = self.B, self.X, topo.n_msgs, topo.P, topo.U
        a.F_ee, a.F_ed, a.Vde = F_ee, F_ed, int(self.phi_en_de.shape[1])
        a.msgs = self.msgs.data_ptr()
        if topo.P:
            a.n_pair_tables = self.pair_tables.shape[0]
            a.pair_tables, a.pair_tab = self.pair_tables.data_ptr(), self.pair_tab.data_ptr()
            a.pair_c_slot, a.pair_r_slot = c.data_ptr(), r.data_ptr()
            a.pair_phi, a.pair_label = self._pair_phi.data_ptr(), self._pair_label.data_ptr()
        if topo.U:
            a.n_unary_tables = self.unary_tables.shape[0]
            a.unary_tables, a.unary_tab = self.unary_tables.data_ptr(), self.unary_tab.data_ptr()
            a.unary_kind, a.unary_obs = self._unary_kind.data_ptr(), self._unary_obs.data_ptr()
            a.unary_label = self._unary_label.data_ptr()
        a.phi_en_en, a.phi_en_en_w1 = self.phi_en_en.data_ptr(), self.phi_en_en_w1.data_ptr()
        a.phi_en_de = self.phi_en_de.data_ptr()
        a.phi_en_en_t, a.phi_en_en_w1_t, a.phi_en_de_t = (p.data_ptr() for p in self._phi_t)
        if getattr(self, 'use_planar', True):
            a.phi_en_en_p, a.phi_en_en_w1_p = (p.data_ptr() for p in self._phi_p)
        a.grad_en_en, a.grad_en_de = out_ee.data_ptr(), out_ed.data_ptr()
        if self.use_approx_beliefs:
            a.flags |= _ffi.GRADIENT_APPROX_BELIEFS
        if getattr(self, 'pair_tables_shared', False) and getattr(self, 'use_shared_gradient', True) and not self.use_approx_beliefs:
            a.flags |= _ffi.GRADIENT_SHARED_PAIR_TABLES
            if getattr(self, '_pair_row_host', None) is not None:
                a.pair_tab_host = self._pair_row_host.ctypes.data
                if not hasattr(self, '_pair_slots_host'):      # host copy of the slot arrays: the X >= 128 gradient then only enqueues
                    self._pair_slots_host = np.ascontiguousarray(np.concatenate([
                        c.cpu().numpy(), r.cpu().numpy(), self._pair_phi.cpu().numpy()]).astype(np.int32))
                a.pair_slots_host = self._pair_slots_host.ctypes.data
            self._derive_unary_rows()
            if getattr(self, '_row_kind', None) is not None and self._row_kind is not False and F_ee == 3 and F_ed == 6:
                # rows [0, done) were written by the caller's potentials launch (mlbp_potentials_job.expect: the trainer's shared
                # pots); what is left -- all rows, or the trainer's private plane-patched ones -- is computed here
                done, n_rows = int(getattr(self, '_uexp_rows_done', 0)), int(self.unary_tables.shape[0])
                if done < n_rows:
                    _ffi.check(_ffi.lib.mlbp_unary_expectations_f64(
                        self.unary_tables[done:].data_ptr(), n_rows - done, self.X, self._row_kind[done:].data_ptr(),
                        self._row_obs[done:].data_ptr(), self._phi_t[0].data_ptr(), self._phi_t[1].data_ptr(), self._phi_t[2].data_ptr(),
                        F_ee, F_ed, int(self.phi_en_de.shape[1]), self._uexp[done:].data_ptr(), _stream_ptr(self.device)))
                a.unary_expect = self._uexp.data_ptr()
        return a

    def gradient(self, out_ee=None, out_ed=None):
        """Per-graph unregularised gradients ([B][F_ee], [B][F_ed]) from the messages in memory:
        FactorGraph.get_unregularized_gradeint (LBP.py:301-320), beliefs fused in.  (sweep(...,
        gradient=(out_ee, out_ed)) produces the same numbers inside the sweep launch.)"""
        F_ee, F_ed = int(self.phi_en_en.shape[2]), int(self.phi_en_de.shape[2])
        if out_ee is None:
            out_ee = torch.empty(self.B, F_ee, dtype=torch.float64, device=self.device)
        if out_ed is None:
            out_ed = torch.empty(self.B, F_ed, dtype=torch.float64, device=self.device)
        a = self._gradient_args(out_ee, out_ed)
        _ffi.check(_ffi.lib.mlbp_gradient_f64(C.byref(a), _stream_ptr(self.device)))
        return out_ee, out_ed

    def sum_rows(self, t, out=None):
        """Column sums of a [rows][cols] device tensor in a fixed order (mlbp_sum_rows_f64)."""
        t2 = t.reshape(t.shape[0], -1)
        if out is None:
            out = torch.empty(t2.shape[1], dtype=torch.float64, device=self.device)
        _ffi.check(_ffi.lib.mlbp_sum_rows_f64(t2.data_ptr(), t2.shape[0], t2.shape[1], out.data_ptr(),
                                              _stream_ptr(self.device)))
        return out


def sweep_groups(batches, roots, init=False, marginals=None, keep_messages=True, gradients=None):
    """One minibatch of mixed graphs: batches[k] (a FactorGraphBatch: one topology, its tables and messages) is swept
    with its own root sequence roots[k] -- the reference draws roots per instance (LBP.py:223-225) and builds a
    different K_n per instance (train_mp.py:257-299).  Same results as batches[k].sweep(roots[k], ...) one by one; when
    every group qualifies, the fast kernel runs them all in ONE launch (mlbp_sweep_groups_f64).
    marginals: None or one [B_k][n_vars_k][X] tensor per group; gradients: None or one (g_en_en, g_en_de) pair per group
    (each group's gradient launch follows its sweeps, as in FactorGraphBatch.sweep(gradient=...))."""
    if len(batches) != len(roots) or not batches:
        raise ValueError('one root sequence per batch')
    dev = batches[0].device
    got = []
    for k, fb in enumerate(batches):
        if fb.device != dev:
            raise ValueError('all groups live on one device')
        fb.sweep(roots[k], init=init, marginals=None if marginals is None else marginals[k], keep_messages=keep_messages,
                 gradient=None if gradients is None else gradients[k], _collect=got)
    n = len(got)
    handles = (C.c_void_p * n)(*[p.handle for p, _, _ in got])
    args = (_ffi.SweepArgs * n)(*[a for _, a, _ in got])
    _ffi.check(_ffi.lib.mlbp_sweep_groups_f64(handles, args, n, _stream_ptr(dev)))
    return [p for p, _, _ in got]
