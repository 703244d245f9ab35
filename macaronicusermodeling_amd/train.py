"""The train_mp.py-shaped outer step on the GPU: potentials -> sweeps -> gradient -> reduce -> theta.

Mirrors `batch_sgd` (train_mp.py:362-402) + `batch_sgd_accumulate` (train_mp.py:405-424) for a
whole shard of instances at once:

    pots   = exp(phi . theta^T)                       train_mp.py:220-255   mlbp_potentials_f64
    fg.initialize(); fg.treelike_inference(3)         train_mp.py:381-382   mlbp_sweep_f64 (init fused)
    g_en_en, g_en_de = fg.return_gradient()           train_mp.py:398       mlbp_gradient_f64
    p = fg.get_posterior_probs()                      train_mp.py:400       mlbp_marginals / log_posterior
    theta += sum_i lr (g_i - reg theta)               train_mp.py:405-424   mlbp_sum_rows_f64 + all-reduce

The reference applies each instance's step as its worker finishes (stale, asynchronous); here every
instance of a step sees the same theta and the steps are summed -- identical to the reference when
all tasks were pickled with the same theta.

Pairwise tables are the two shared pots (`pot_en_en`, `pot_en_en_w1`) referenced through the
table-index indirection; a unary factor's table is a COLUMN of a pot (LBP.py:702-703), read as a
row of the transposed pot the potentials kernel also writes -- no per-instance copies.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _ffi, dist as mdist
from .batch import FactorGraphBatch, _stream_ptr
from .topology import GraphTopology


class UserGraphTrainer:
    def __init__(self, spec, var_labels, unary_obs, phi_en_en, phi_en_en_w1, phi_en_de, theta_en_en, theta_en_de,
                 device='cuda:0', sweeps=3, roots=None, planes=None, domains=None, theta_dom_en_en=None,
                 theta_dom_en_de=None, skip_unchanged=False, shared=None, features_of=None):
        """spec: a 'trainmp'-style spec (tests/golden/cases.py: factors carry factor_type / gap);
        var_labels [B][n_vars], unary_obs [B][U]: this rank's shard of instances.
        planes: optional per-instance sparse feature planes, a list (one entry per instance) of
        {(i, j, k): value} -- cell (i, j) of phi_en_de[:, :, k] for this instance only
        (train_mp.py:178-217 writes k = 2 'correct', 3 'full_history', 4 'hit_history').
        domains [B] + theta_dom_en_en [D][F_ee] + theta_dom_en_de [D][F_ed]: --user_adapt / --experience_adapt
        (train_mp.py:162-171, 226-247): instance i builds its potentials from theta_dom[domains[i]] INSTEAD of the
        global theta; the global theta still receives every instance's step.  Instances of one domain should be
        contiguous (groups of 16 consecutive graphs that share their tables run on the matrix cores).
        shared: a _SharedTables (the bucket trainers of one TiDirTrainer): the pots, their transposed rows, every bucket's private
        rows and the per-instance outputs then live in arrays the SET owns -- one potentials launch, one patch launch and one
        statistics launch per step for all sentence shapes instead of one each per shape; finish_shared() completes the
        construction once the set has sized them.
        skip_unchanged (off by default, like the C ABI: the default executes every update of the reference's schedule): the
        sweeps drop the updates of the root sequence that would recompute a message from unchanged inputs
        (MLBP_SWEEP_SKIP_UNCHANGED, include/mlbp.h) -- a third of a three-root user graph's contractions; statistics equal
        the full schedule's to rounding (tests/test_gpu_gradient.py)."""
        self.spec = spec
        self.topo = topo = GraphTopology.from_spec(spec)
        by_id = {f['id']: f for f in spec['factors']}
        X = spec['X']
        self.device = torch.device(device)
        B = int(np.asarray(var_labels).shape[0])
        self.batch = fb = FactorGraphBatch(topo, X, B, device=self.device)
        fb.skip_unchanged = bool(skip_unchanged)
        pair_phi, unary_kind = [], []
        for j in topo.pair_factors:
            f = by_id[topo.factor_ids[j]]
            if f['factor_type'] != 'en_en' or f['gap'] < 1:
                raise BaseException('only 2 kinds of distances are supported ...')      # LBP.py:463
            pair_phi.append(0 if f['gap'] > 1 else 1)
        for j in topo.unary_factors:
            f = by_id[topo.factor_ids[j]]
            if f['factor_type'] == 'en_de':
                unary_kind.append(2)
            elif f['factor_type'] == 'en_en' and f['gap'] >= 1:
                unary_kind.append(0 if f['gap'] > 1 else 1)
            else:
                raise BaseException('only two kinds of potentials are supported...')    # LBP.py:467
        # features_of: another trainer given the SAME feature tensors -- its device copies are used (the buckets of one TiDirTrainer)
        fb.set_features(phi_en_en, phi_en_en_w1, phi_en_de, pair_phi, unary_kind, share_with=features_of.batch if features_of is not None else None)
        fb.set_observations(var_labels, unary_obs)
        self.Vde = int(fb.phi_en_de.shape[1])
        self.F_ee, self.F_ed = int(fb.phi_en_en.shape[2]), int(fb.phi_en_de.shape[2])
        dev = self.device
        # theta may be passed as device tensors so that several shape buckets share one parameter vector
        self.theta_en_en = theta_en_en if isinstance(theta_en_en, torch.Tensor) else \
            torch.as_tensor(np.asarray(theta_en_en, dtype=np.float64).reshape(-1)).to(dev)
        self.theta_en_de = theta_en_de if isinstance(theta_en_de, torch.Tensor) else \
            torch.as_tensor(np.asarray(theta_en_de, dtype=np.float64).reshape(-1)).to(dev)
        # pots: [pot_en_en, pot_en_en_w1] as pairwise tables; their transposes + pot_en_de^T as unary rows
        as_theta = lambda t: t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t, dtype=np.float64)).to(dev)   # noqa: E731
        self.n_dom = 0
        self._dom_host = np.zeros(B, dtype=np.int64)
        if domains is not None:
            self.theta_dom_en_en, self.theta_dom_en_de = as_theta(theta_dom_en_en), as_theta(theta_dom_en_de)
            self.n_dom = int(self.theta_dom_en_en.shape[0])
            self._dom_host = np.asarray(domains, dtype=np.int64).reshape(B)
            if self.n_dom < 1 or self._dom_host.min() < 0 or self._dom_host.max() >= self.n_dom:
                raise IndexError('domain index out of range')
            if tuple(self.theta_dom_en_en.shape) != (self.n_dom, self.F_ee) or tuple(self.theta_dom_en_de.shape) != (self.n_dom, self.F_ed):
                raise ValueError('theta_dom_* must be [D][F]')
            self._dom = torch.from_numpy(self._dom_host.astype(np.int32)).to(dev)
        nd = max(self.n_dom, 1)
        self.rows_per_dom = 2 * X + self.Vde
        obs_np = np.asarray(unary_obs, dtype=np.int64).reshape(B, topo.U)
        self._plan_patches(planes, unary_kind, obs_np, np.asarray(var_labels, dtype=np.int64).reshape(B, topo.n_vars))
        self.n_shared_rows = self.rows_per_dom * nd
        self._pair_phi, self._unary_kind_list, self._obs_np = pair_phi, unary_kind, obs_np
        self.sweeps = int(sweeps)
        self.roots = list(roots) if roots is not None else [topo.var_ids[i % topo.n_vars] for i in range(self.sweeps)]
        fb.is_loopy = topo.has_loops(self.roots[0])
        self.n_sweeps_run = self.sweeps if fb.is_loopy else 1                            # LBP.py:219
        self.shared = shared
        if shared is None:
            self.pair_tables = torch.empty(2 * nd, X, X, dtype=torch.float64, device=dev)
            self.unary_tables = torch.empty(self.n_shared_rows + self.n_priv, X, dtype=torch.float64, device=dev)
            self._finish(priv_row0=self.n_shared_rows, g_ee=torch.empty(B, self.F_ee, dtype=torch.float64, device=dev),
                         g_ed=torch.empty(B, self.F_ed, dtype=torch.float64, device=dev))

    def finish_shared(self, priv_row0, g_ee, g_ed):
        """Second half of the construction under a _SharedTables: the set's pots and rows, this bucket's private rows starting
        at row priv_row0 of the set's row array, its per-instance gradient rows as views of the set's arrays."""
        self.pair_tables, self.unary_tables = self.shared.pair_tables, self.shared.unary_tables
        self._finish(priv_row0, g_ee, g_ed)

    def _finish(self, priv_row0, g_ee, g_ed):
        fb, topo, dev = self.batch, self.topo, self.device
        X, B = self.spec['X'], fb.B
        self.priv_row0 = int(priv_row0)
        pair_phi, unary_kind = self._pair_phi, self._unary_kind_list
        fb.pair_tables = self.pair_tables
        ptab = np.tile(np.array(pair_phi or [0], dtype=np.int64), (B, 1)) + 2 * self._dom_host[:, None]
        fb.pair_tab = torch.from_numpy(ptab.astype(np.int32)).to(dev)
        fb.pair_tables_shared = bool(topo.P)      # every instance reads the two pots: the MFMA kernels apply
        # one row for every instance (no per-domain pots): at X >= 128 the sweeps can run as batched MFMA contractions
        fb._pair_row_host = np.ascontiguousarray(ptab[0], dtype=np.int32) if (topo.P and not self.n_dom) else None
        obs = self._obs_np
        base = np.array([0, X, 2 * X], dtype=np.int64)[np.array(unary_kind, dtype=np.int64)] if topo.U else np.zeros(0)
        fb.unary_tables = self.unary_tables
        utab = obs + base[None, :] + self.rows_per_dom * self._dom_host[:, None]
        for r, (b_i, u) in enumerate(self._priv_rows):          # patched factors read their private row
            utab[b_i, u] = self.priv_row0 + r
        fb.unary_tab = torch.from_numpy(utab.astype(np.int32)).to(dev)
        self._g_ee, self._g_ed = g_ee, g_ed
        self._marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
        self._lp = torch.empty(B, dtype=torch.float64, device=dev)
        n_stat = self.F_ee + self.F_ed + 2
        self._rows = torch.empty(B, n_stat, dtype=torch.float64, device=dev)
        # one buffer for the single all-reduce of a step: [global statistics | per-domain statistics [D][n_stat]]
        self.stats_all = torch.zeros(n_stat * (1 + self.n_dom), dtype=torch.float64, device=dev)
        self.stats = self.stats_all[:n_stat]
        self.stats_dom = self.stats_all[n_stat:].view(self.n_dom, n_stat) if self.n_dom else None
        # X = 64, shared pots: the potentials launch also writes every shared row's expected features (the gradient's gather
        # table); only the private (plane-patched) rows still go through mlbp_unary_expectations_f64
        self._expect_in_potentials = False
        if X == 64 and topo.U and self.F_ee == 3 and self.F_ed == 6:
            if self.shared is not None:                 # the set states the rows' (kind, column) and fills every row's expectations
                fb._row_kind, fb._row_obs, fb._uexp = self.shared.row_kind, self.shared.row_obs, self.shared.uexp
                fb._uexp_rows_done = int(self.shared.unary_tables.shape[0])
                self._expect_in_potentials = True
            else:
                fb._derive_unary_rows()
                if getattr(fb, '_row_kind', None) is not None and fb._row_kind is not False:
                    self._expect_in_potentials = True
                    fb._uexp_rows_done = self.n_shared_rows

    def _plan_patches(self, planes, unary_kind, obs, labels):
        """Host-side integer work: which (instance, en_de factor) pairs see a plane cell in their
        observed column, and the CSR item lists the two patch kernels consume."""
        topo, X = self.topo, self.spec['X']
        rows, off, ix, ik, iv, rgraph, rlabel, rbase = [], [0], [], [], [], [], [], []
        if planes is not None:
            ed_slots = [u for u in range(topo.U) if unary_kind[u] == 2]
            for b_i, cells in enumerate(planes):
                if not cells:
                    continue
                for u in ed_slots:
                    col = int(obs[b_i, u])
                    items = [(i, k, v) for (i, j, k), v in sorted(cells.items()) if j == col and v != 0.0]
                    if not items:
                        continue
                    for i, k, v in items:
                        if not (0 <= i < X and 0 <= k < self.F_ed):
                            raise IndexError('feature-plane cell out of range')
                        ix.append(i); ik.append(k); iv.append(float(v))
                    rows.append((b_i, u)); off.append(len(ix))
                    rgraph.append(b_i); rbase.append(2 * X + col + self.rows_per_dom * int(self._dom_host[b_i]))
                    rlabel.append(int(labels[b_i, topo.fac_var[2 * topo.unary_factors[u]]]))
        self._priv_rows, self.n_priv = rows, len(rows)
        self._priv_cols = [b - 2 * X - self.rows_per_dom * int(self._dom_host[g_i]) for b, g_i in zip(rbase, rgraph)]   # observed column of each private row
        # private rows of one domain must be contiguous (their exponent uses that domain's theta): instances come
        # grouped by domain, so the row list already is; record the ranges
        row_dom = [int(self._dom_host[b_i]) for b_i, _ in rows]
        if any(row_dom[i] > row_dom[i + 1] for i in range(len(row_dom) - 1)):
            raise ValueError('instances with feature planes must be grouped by domain')
        self._priv_dom_ranges = []
        for d in sorted(set(row_dom)):
            lo = row_dom.index(d)
            self._priv_dom_ranges.append((d, lo, lo + row_dom.count(d)))
        self._p_host = (off, ix, ik, iv, rgraph, rlabel, rbase)      # (a _SharedTables joins the sets' plans into one launch)
        if self.n_priv:
            dev = self.device
            as_i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)          # noqa: E731
            self._p_off, self._p_x, self._p_k = as_i32(off), as_i32(ix), as_i32(ik)
            self._p_val = torch.tensor(iv, dtype=torch.float64, device=dev)
            self._p_graph, self._p_label, self._p_base = as_i32(rgraph), as_i32(rlabel), as_i32(rbase)

    def _patch_tables(self):
        if self.n_priv:
            priv = self.unary_tables[self.priv_row0:]
            for d, lo, hi in self._priv_dom_ranges:          # one launch per domain present (its theta in the exponent)
                theta = self.theta_dom_en_de[d] if self.n_dom else self.theta_en_de
                _ffi.check(_ffi.lib.mlbp_patch_unary_tables_f64(
                    self.unary_tables.data_ptr(), self._p_base[lo:].data_ptr(), self._p_off[lo:].data_ptr(), self._p_x.data_ptr(),
                    self._p_k.data_ptr(), self._p_val.data_ptr(), theta.data_ptr(), hi - lo, self.spec['X'],
                    priv[lo:].data_ptr(), _stream_ptr(self.device)))

    def _patch_gradient(self):
        if self.n_priv:
            priv = self.unary_tables[self.priv_row0:]
            _ffi.check(_ffi.lib.mlbp_patch_gradient_f64(
                priv.data_ptr(), self._p_off.data_ptr(), self._p_x.data_ptr(), self._p_k.data_ptr(), self._p_val.data_ptr(),
                self._p_graph.data_ptr(), self._p_label.data_ptr(), self.n_priv, self.spec['X'], self.F_ed,
                self._g_ed.data_ptr(), _stream_ptr(self.device)))

    def build_potentials(self):
        if self.shared is not None:
            return self.shared.build([self])
        return self._build_potentials_into(self.pair_tables, self.unary_tables, self.batch._uexp if self._expect_in_potentials else None)

    def _build_potentials_into(self, pt, ut, ue, patch=True):
        fb, X, st = self.batch, self.spec['X'], _stream_ptr(self.device)
        # all three pots, for the global theta or for every domain's (its theta REPLACES the global one,
        # train_mp.py:226-247), in ONE launch: pot_en_en / pot_en_en_w1 as pairwise tables and transposed rows, pot_en_de
        # as transposed rows only
        nd = max(self.n_dom, 1)
        t_ee = self.theta_dom_en_en if self.n_dom else self.theta_en_en
        t_ed = self.theta_dom_en_de if self.n_dom else self.theta_en_de
        # ... and, X = 64 (ue given), every shared row's expected features (what the gradient gathers per unary factor,
        # mlbp_unary_expectations_f64) out of the same launch: a row of pot^T IS a unary factor's table
        jobs = (_ffi.PotentialsJob * 3)()
        for j, (phi, th, F, cols, pot, pot_t, row0) in enumerate((
                (fb.phi_en_en, t_ee, self.F_ee, X, pt[0].data_ptr(), ut[0:X].data_ptr(), 0),
                (fb.phi_en_en_w1, t_ee, self.F_ee, X, pt[1].data_ptr(), ut[X:2 * X].data_ptr(), X),
                (fb.phi_en_de, t_ed, self.F_ed, self.Vde, None, ut[2 * X:].data_ptr(), 2 * X))):
            jobs[j].phi, jobs[j].theta, jobs[j].pot, jobs[j].pot_t = phi.data_ptr(), th.data_ptr(), pot, pot_t
            jobs[j].theta_stride, jobs[j].pot_stride, jobs[j].pot_t_stride = F, 2 * X * X, self.rows_per_dom * X
            jobs[j].rows, jobs[j].cols, jobs[j].F = X, cols, F
            if ue is not None:
                jobs[j].expect, jobs[j].expect_stride = ue[row0:].data_ptr(), self.rows_per_dom * 8
        _ffi.check(_ffi.lib.mlbp_potentials_multi_f64(jobs, 3, nd, st))
        if patch:
            self._patch_tables()

    def capture(self):
        """Records local_statistics() into a HIP graph (torch.cuda.CUDAGraph over the library's stream-ordered
        launches): the step is a dozen short kernels, so replaying one graph instead of issuing them from Python
        removes the launch-bound host time.  theta is read from the same device tensors at every replay.  Call
        after one eager local_statistics() (first-use allocations and attribute calls must not be captured)."""
        self._graph = None
        self._local_statistics_eager()               # warm-up on the current stream
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._local_statistics_eager()
        self._graph = g
        return self

    def local_statistics(self):
        """Runs inference on this rank's shard and returns the fused statistics buffer (device):
        [sum_i grad_en_en (F_ee) | sum_i grad_en_de (F_ed) | sum_i log-posterior | instance count]."""
        if getattr(self, '_graph', None) is not None:
            self._graph.replay()
            return self.stats_all if self.n_dom else self.stats
        return self._local_statistics_eager()

    def _sweep_with_gradient(self):
        self.batch.sweep(self.roots[:self.n_sweeps_run], init=True, marginals=self._marg, gradient=(self._g_ee, self._g_ed),
                         keep_messages=False)

    def _local_statistics_eager(self, select=None, potentials=True):
        if potentials:
            self.build_potentials()
        self._sweep_with_gradient()
        return self._statistics_after_sweep(select=select)

    def _statistics_after_sweep(self, gradient_from_messages=False, select=None):
        """Everything of the step behind the sweeps: gradient (when the sweep launch did not produce it: from the
        messages in memory), the per-instance plane shares, log-posteriors, the batch sums.
        select: optional device bool [B] -- only THESE instances enter the sums (a minibatch of a resident shard: every instance
        is evaluated, the statistics of the others are left out; TiDirTrainer's masked minibatches)."""
        fb = self.batch
        if gradient_from_messages:
            fb.gradient(self._g_ee, self._g_ed)
        self._patch_gradient()
        if select is not None:
            return self._selected_statistics(select)
        if not self.n_dom:          # [sum g_ee | sum g_ed | sum log-posterior | count]: log-posteriors and sums in one launch
            _ffi.check(_ffi.lib.mlbp_step_statistics_f64(self._g_ee.data_ptr(), self.F_ee, self._g_ed.data_ptr(), self.F_ed,
                                                         self._marg.data_ptr(), fb._labels.data_ptr(), self.topo.n_vars, fb.X, fb.B,
                                                         self._lp.data_ptr(), self.stats.data_ptr(), _stream_ptr(self.device)))
            return self.stats
        _ffi.check(_ffi.lib.mlbp_log_posterior_f64(self._marg.data_ptr(), fb._labels.data_ptr(), fb.B, self.topo.n_vars,
                                                   fb.X, self._lp.data_ptr(), _stream_ptr(self.device)))
        r = self._rows
        r[:, :self.F_ee] = self._g_ee
        r[:, self.F_ee:self.F_ee + self.F_ed] = self._g_ed
        r[:, -2] = self._lp
        r[:, -1] = 1.0
        fb.sum_rows(r, out=self.stats)
        if self.n_dom:
            _ffi.check(_ffi.lib.mlbp_segment_sum_rows_f64(r.data_ptr(), fb.B, r.shape[1], self._dom.data_ptr(), self.n_dom,
                                                          self.stats_dom.data_ptr(), _stream_ptr(self.device)))
            return self.stats_all
        return self.stats

    def _selected_statistics(self, select):
        """The statistics buffer over the instances with select[b] set: per-instance rows [g_ee | g_ed | log-posterior | 1],
        summed by segment -- the instance's domain when selected, a dump segment otherwise (mlbp_segment_sum_rows_f64, fixed
        order)."""
        fb = self.batch
        st = _stream_ptr(self.device)
        _ffi.check(_ffi.lib.mlbp_log_posterior_f64(self._marg.data_ptr(), fb._labels.data_ptr(), fb.B, self.topo.n_vars,
                                                   fb.X, self._lp.data_ptr(), st))
        r = self._rows
        r[:, :self.F_ee] = self._g_ee
        r[:, self.F_ee:self.F_ee + self.F_ed] = self._g_ed
        r[:, -2] = self._lp
        r[:, -1] = 1.0
        nd = max(self.n_dom, 1)
        if getattr(self, '_segsum', None) is None:
            self._segsum = torch.zeros(nd + 1, r.shape[1], dtype=torch.float64, device=self.device)
            self._seg = torch.empty(fb.B, dtype=torch.int32, device=self.device)
            self._dump = torch.full((fb.B,), nd, dtype=torch.int32, device=self.device)
            self._zero_seg = torch.zeros(fb.B, dtype=torch.int32, device=self.device)
        torch.where(select, self._dom if self.n_dom else self._zero_seg, self._dump, out=self._seg)
        _ffi.check(_ffi.lib.mlbp_segment_sum_rows_f64(r.data_ptr(), fb.B, r.shape[1], self._seg.data_ptr(), nd + 1,
                                                      self._segsum.data_ptr(), st))
        if self.n_dom:
            self.stats_dom.copy_(self._segsum[:nd])
            torch.sum(self._segsum[:nd], dim=0, out=self.stats)
            return self.stats_all
        self.stats.copy_(self._segsum[0])
        return self.stats

    def step(self, learning_rate, reg_param, reg_param_ua_scale=1.0):
        """One synchronous optimisation step over ALL ranks' shards: returns (mean log-posterior,
        theta_en_en, theta_en_de).  reg_param is FactorGraph.regularization_param, i.e. the reference's
        `--reg_param / N` (train_mp.py:160); with domains the per-domain thetas take their own instances' steps
        with the regulariser scaled by reg_param_ua_scale (train_mp.py:384-396, 413-415)."""
        stats = mdist.all_reduce_sum_(self.local_statistics())
        n_stat = self.F_ee + self.F_ed + 2
        apply_update(self.theta_en_en, self.theta_en_de, stats[:n_stat], self.F_ee, self.F_ed, learning_rate, reg_param)
        if self.n_dom:
            apply_domain_update(self.theta_dom_en_en, self.theta_dom_en_de, stats[n_stat:].view(self.n_dom, n_stat),
                                self.F_ee, self.F_ed, learning_rate, reg_param * reg_param_ua_scale)
        return float(stats[n_stat - 2].item() / stats[n_stat - 1].item()), self.theta_en_en, self.theta_en_de

    def predict(self, top=50, with_logs=False):
        """batch_predictions for the shard (train_mp.py:310-343): runs inference and returns
        (log-posterior per instance [B] (host), top-`top` word indices per predicted variable
        [B][n_vars][top] in descending probability (device top-K), their log-probabilities, and the
        precision counts (p@0, p@25, p@50, total) of FactorGraph.get_precision_counts, LBP.py:80-106)."""
        fb, topo = self.batch, self.topo
        self.build_potentials()
        fb.sweep(self.roots[:self.n_sweeps_run], init=True, marginals=self._marg, keep_messages=False)
        st = _stream_ptr(self.device)
        _ffi.check(_ffi.lib.mlbp_log_posterior_f64(self._marg.data_ptr(), fb._labels.data_ptr(), fb.B, topo.n_vars, fb.X,
                                                   self._lp.data_ptr(), st))
        idx = torch.empty(fb.B, topo.n_vars, top, dtype=torch.int32, device=self.device)
        _ffi.check(_ffi.lib.mlbp_topk_rows_f64(self._marg.data_ptr(), fb.B * topo.n_vars, fb.X, top, idx.data_ptr(), st))
        logm = torch.empty_like(self._marg)
        _ffi.check(_ffi.lib.mlbp_log_f64(self._marg.data_ptr(), logm.data_ptr(), self._marg.numel(), st))
        idx_h = idx.cpu().numpy().astype(np.int64)
        logs = np.take_along_axis(logm.cpu().numpy(), idx_h, axis=2)
        labels = fb._labels.cpu().numpy()
        # integer work: rank of the user's label among the top words of every en_de factor's variable
        by_id = {f['id']: f for f in self.spec['factors']}
        ed_vars = [int(topo.fac_var[2 * j]) for j in topo.unary_factors if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de']
        hit = idx_h[:, ed_vars, :] == labels[:, ed_vars, None]
        rank = np.where(hit.any(-1), hit.argmax(-1), 10 ** 6)
        counts = (int((rank == 0).sum()), int((rank < 26).sum()), int((rank < 51).sum()), int(rank.size))
        if with_logs:       # + every variable's log-marginal vector and the log-probability of the user's label (the text formats)
            logm_h = logm.cpu().numpy()
            return self._lp.cpu().numpy(), idx_h, logs, counts, logm_h, np.take_along_axis(logm_h, labels[:, :, None], axis=2)[:, :, 0]
        return self._lp.cpu().numpy(), idx_h, logs, counts


def apply_update(theta_en_en, theta_en_de, stats, F_ee, F_ed, learning_rate, reg_param):
    """theta += sum_i lr (g_i - reg theta) = lr (sum_i g_i - n reg theta): the sum of the per-instance
    steps `return_gradient` produces (LBP.py:293-299, 322-327) as `batch_sgd_accumulate` adds them
    (train_mp.py:419-424).  Works on CPU or device tensors (it is part of the gloo tests)."""
    n = stats[-1]
    theta_en_en += learning_rate * (stats[:F_ee] - n * reg_param * theta_en_en)
    theta_en_de += learning_rate * (stats[F_ee:F_ee + F_ed] - n * reg_param * theta_en_de)
    return theta_en_en, theta_en_de


def apply_domain_update(theta_dom_en_en, theta_dom_en_de, stats_dom, F_ee, F_ed, learning_rate, reg_param):
    """theta_d += lr (sum_{i in d} g_i - n_d reg theta_d): the per-domain half of batch_sgd_accumulate
    (train_mp.py:384-396, 413-415; `reg_param` already carries reg_param_ua_scale).  stats_dom: [D][F_ee+F_ed+2]."""
    n = stats_dom[:, -1:]
    theta_dom_en_en += learning_rate * (stats_dom[:, :F_ee] - n * reg_param * theta_dom_en_en)
    theta_dom_en_de += learning_rate * (stats_dom[:, F_ee:F_ee + F_ed] - n * reg_param * theta_dom_en_de)
    return theta_dom_en_en, theta_dom_en_de


class _SharedTables:
    """What the bucket trainers of one set have in common, held ONCE: the pots under the global theta (or every domain's) as
    pairwise tables and as transposed rows (train_mp.py:220-255 builds them per instance; they depend on theta alone), behind
    the shared rows every bucket's private plane-patched rows, the rows' expected features, and the per-instance gradient rows
    of all buckets.  One potentials launch and one expectations launch per step serve every sentence shape."""

    def __init__(self, trainers):
        t0 = trainers[0]
        self.trainers = list(trainers)
        dev, X = t0.device, t0.spec['X']
        nd = max(t0.n_dom, 1)
        self.n_shared_rows = t0.n_shared_rows
        n_priv = sum(t.n_priv for t in trainers)
        self.pair_tables = torch.empty(2 * nd, X, X, dtype=torch.float64, device=dev)
        self.unary_tables = torch.empty(self.n_shared_rows + n_priv, X, dtype=torch.float64, device=dev)
        n_inst = sum(t.batch.B for t in trainers)
        self.g_ee = torch.empty(n_inst, t0.F_ee, dtype=torch.float64, device=dev)
        self.g_ed = torch.empty(n_inst, t0.F_ed, dtype=torch.float64, device=dev)
        # every row's (phi selector, observed column): shared rows by their place in the layout, private rows their base row's
        rk = np.zeros(self.n_shared_rows + n_priv, dtype=np.int32)
        ro = np.zeros(self.n_shared_rows + n_priv, dtype=np.int32)
        for d in range(nd):
            o = d * t0.rows_per_dom
            rk[o + X:o + 2 * X] = 1; rk[o + 2 * X:o + t0.rows_per_dom] = 2
            ro[o:o + X] = np.arange(X); ro[o + X:o + 2 * X] = np.arange(X); ro[o + 2 * X:o + t0.rows_per_dom] = np.arange(t0.Vde)
        row, inst = self.n_shared_rows, 0
        self._plans = []
        for t in trainers:
            rk[row:row + t.n_priv] = 2
            ro[row:row + t.n_priv] = np.asarray(t._priv_cols, dtype=np.int32).reshape(-1)
            self._plans.append((row, inst))
            row += t.n_priv; inst += t.batch.B
        self.row_kind = torch.from_numpy(rk).to(dev)
        self.row_obs = torch.from_numpy(ro).to(dev)
        self.uexp = torch.zeros(self.n_shared_rows + n_priv, 8, dtype=torch.float64, device=dev)
        for t, (row, inst) in zip(trainers, self._plans):
            t.shared = self
            t.finish_shared(row, self.g_ee[inst:inst + t.batch.B], self.g_ed[inst:inst + t.batch.B])
        self.n_inst, self.n_priv, self.n_dom = n_inst, n_priv, t0.n_dom
        self.F_ee, self.F_ed, self.X = t0.F_ee, t0.F_ed, X
        # the buckets' patch plans joined: one mlbp_patch_unary_tables_f64 (one theta: no per-domain thetas) and one
        # mlbp_patch_gradient_f64 launch for every sentence shape
        off, ix, ik, iv, rg, rl, rb = [0], [], [], [], [], [], []
        for t, (row, inst) in zip(trainers, self._plans):
            o, x, k, v, g, l, b = t._p_host
            off += [len(ix) + e for e in o[1:]]
            ix += x; ik += k; iv += v; rl += l; rb += b
            rg += [inst + gi for gi in g]
        if n_priv:
            as_i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)          # noqa: E731
            self._p_off, self._p_x, self._p_k = as_i32(off), as_i32(ix), as_i32(ik)
            self._p_val = torch.tensor(iv, dtype=torch.float64, device=dev)
            self._p_graph, self._p_label, self._p_base = as_i32(rg), as_i32(rl), as_i32(rb)
        # per-instance outputs of the whole set and the group table of mlbp_log_posterior_groups_f64
        self.lp = torch.empty(n_inst, dtype=torch.float64, device=dev)
        n_stat = self.F_ee + self.F_ed + 2
        self.rows = torch.empty(n_inst, n_stat, dtype=torch.float64, device=dev)
        self.stats_all = torch.zeros(n_stat * (1 + self.n_dom), dtype=torch.float64, device=dev)
        gt = np.zeros(len(trainers), dtype=[('m', '<u8'), ('l', '<u8'), ('nv', '<i4'), ('B', '<i4'), ('start', '<i8')])
        for i, (t, (row, inst)) in enumerate(zip(trainers, self._plans)):
            gt[i] = (t._marg.data_ptr(), t.batch._labels.data_ptr(), t.topo.n_vars, t.batch.B, inst)
        self._groups = torch.from_numpy(gt.view(np.uint8).reshape(-1).copy()).to(dev)
        nd = max(self.n_dom, 1)
        self._segsum = torch.zeros(nd + 1, n_stat, dtype=torch.float64, device=dev)
        self._seg = torch.empty(n_inst, dtype=torch.int32, device=dev)
        self._dump = torch.full((n_inst,), nd, dtype=torch.int32, device=dev)
        self._seg0 = torch.zeros(n_inst, dtype=torch.int32, device=dev)
        if self.n_dom:
            self._dom = torch.cat([t._dom for t in trainers])

    def patch_gradient(self):
        if self.n_priv:
            priv = self.unary_tables[self.n_shared_rows:]
            _ffi.check(_ffi.lib.mlbp_patch_gradient_f64(
                priv.data_ptr(), self._p_off.data_ptr(), self._p_x.data_ptr(), self._p_k.data_ptr(), self._p_val.data_ptr(),
                self._p_graph.data_ptr(), self._p_label.data_ptr(), self.n_priv, self.X, self.F_ed,
                self.g_ed.data_ptr(), _stream_ptr(self.trainers[0].device)))

    def statistics(self, select=None):
        """[sum g_ee | sum g_ed | sum log-posterior | count] (then the per-domain rows) over every instance of the set -- or over
        the instances select = (key, value) picks: key device int32 [n_inst], value device int32 [1], instance i counts when
        key[i] == value (a masked minibatch) -- behind the sweeps of all buckets: the planes' gradient share, the
        log-posteriors of every group and the sums, one launch each."""
        st = _stream_ptr(self.trainers[0].device)
        self.patch_gradient()
        _ffi.check(_ffi.lib.mlbp_log_posterior_groups_f64(self._groups.data_ptr(), len(self.trainers), self.n_inst, self.X, self.lp.data_ptr(), st))
        n_stat = self.F_ee + self.F_ed + 2
        if not self.n_dom:
            key, value = (None, None) if select is None else (select[0].data_ptr(), select[1].data_ptr())
            _ffi.check(_ffi.lib.mlbp_select_sum_rows_cat_f64(self.g_ee.data_ptr(), self.F_ee, self.g_ed.data_ptr(), self.F_ed, self.lp.data_ptr(), 1,
                                                             self.n_inst, key, value, 1, self.stats_all.data_ptr(), st))
            return self.stats_all
        r = self.rows
        r[:, :self.F_ee] = self.g_ee
        r[:, self.F_ee:self.F_ee + self.F_ed] = self.g_ed
        r[:, -2] = self.lp
        r[:, -1] = 1.0
        nd = max(self.n_dom, 1)
        code = self._dom if self.n_dom else self._seg0
        if select is not None:
            torch.where(select[0] == select[1], code, self._dump, out=self._seg)
            code = self._seg
        _ffi.check(_ffi.lib.mlbp_segment_sum_rows_f64(r.data_ptr(), self.n_inst, n_stat, code.data_ptr(), nd + 1, self._segsum.data_ptr(), st))
        if self.n_dom:
            self.stats_all[n_stat:].view(nd, n_stat).copy_(self._segsum[:nd])
            torch.sum(self._segsum[:nd], dim=0, out=self.stats_all[:n_stat])
        else:
            self.stats_all.copy_(self._segsum[0])
        return self.stats_all

    def build(self, trainers=None):
        """The pots (and the shared rows' expected features) under the current thetas, then the private rows of `trainers`
        (default: all) and THEIR expected features."""
        t0 = self.trainers[0]
        expect = t0._expect_in_potentials
        t0._build_potentials_into(self.pair_tables, self.unary_tables, self.uexp if expect else None, patch=False)
        trs = self.trainers if trainers is None else trainers
        if trainers is None and self.n_priv and not self.n_dom:      # every private row of every shape in one launch
            priv = self.unary_tables[self.n_shared_rows:]
            _ffi.check(_ffi.lib.mlbp_patch_unary_tables_f64(
                self.unary_tables.data_ptr(), self._p_base.data_ptr(), self._p_off.data_ptr(), self._p_x.data_ptr(), self._p_k.data_ptr(),
                self._p_val.data_ptr(), t0.theta_en_de.data_ptr(), self.n_priv, self.X, priv.data_ptr(), _stream_ptr(t0.device)))
        else:
            for t in trs:
                t._patch_tables()
        if not expect:
            return
        spans = [(self.n_shared_rows, int(self.unary_tables.shape[0]))] if trainers is None else [(t.priv_row0, t.priv_row0 + t.n_priv) for t in trs]
        fb = t0.batch
        for lo, hi in spans:
            if hi > lo:
                _ffi.check(_ffi.lib.mlbp_unary_expectations_f64(
                    self.unary_tables[lo:].data_ptr(), hi - lo, fb.X, self.row_kind[lo:].data_ptr(), self.row_obs[lo:].data_ptr(),
                    fb._phi_t[0].data_ptr(), fb._phi_t[1].data_ptr(), fb._phi_t[2].data_ptr(), t0.F_ee, t0.F_ed, t0.Vde,
                    self.uexp[lo:].data_ptr(), _stream_ptr(t0.device)))


class _BucketSet:
    """The bucket trainers (one UserGraphTrainer per sentence shape) of a list of instances, sharing the owner's thetas."""

    def __init__(self, owner, instances):
        from . import tidir
        self.buckets = tidir.bucket_instances(instances, owner.en, owner.de)
        self.trainers = {}
        o = owner
        dom_of = (lambda r: str(r['user_id'])) if o.adapt == 'user' else (lambda r: str(r['n_seen']))
        dom_index = {d: i for i, d in enumerate(o.domains)}
        for key, b in sorted(self.buckets.items()):
            roots = [key[1][i % len(key[1])] for i in range(o.sweeps)]
            extra = {}
            if o.adapt:   # group the bucket's instances by domain: groups of 16 graphs then share their tables
                dom = np.array([dom_index[dom_of(r)] for r in b['rows']], dtype=np.int64)
                order = np.argsort(dom, kind='stable')
                b['rows'] = [b['rows'][i] for i in order]
                b['var_labels'], b['unary_obs'] = b['var_labels'][order], b['unary_obs'][order]
                extra = dict(domains=dom[order], theta_dom_en_en=o.theta_dom_en_en, theta_dom_en_de=o.theta_dom_en_de)
            planes = None
            if o.use_planes:
                planes = []
                for r in b['rows']:
                    cells = {}
                    for name, k in o.plane_features.items():
                        for i, j, v in r['planes'][name]:
                            cells[(i, j, k)] = cells.get((i, j, k), 0.0) + v      # the reference accumulates (+=)
                    planes.append(cells)
            self.trainers[key] = UserGraphTrainer(b['spec'], b['var_labels'], b['unary_obs'], o.phi_ee, o.phi_w1, o.phi_ed_t,
                                                  o.theta_en_en, o.theta_en_de, device=o.device, sweeps=o.sweeps, roots=roots,
                                                  planes=planes, skip_unchanged=o.skip_unchanged, shared=True,
                                                  features_of=next(iter(self.trainers.values()), None), **extra)
        self.shared = _SharedTables(list(self.trainers.values())) if self.trainers else None

    def statistics_into(self, stats, grouped_sweeps, select=None):
        """Adds the set's statistics to `stats`.  grouped_sweeps: the sweeps of ALL sentence shapes with pairwise factors
        in one launch sequence (batch.sweep_groups -> mlbp_sweep_groups_f64: every bucket its own topology and roots;
        the shared-table kernels take a group table) instead of one launch sequence per bucket; 'auto' groups whenever
        two or more buckets qualify.  select: optional device bool over the set's instances (bucket order): only the selected
        ones enter the sums (masked minibatches).  One potentials launch in front of the sweeps and three launches behind them
        (planes' gradient share, log-posteriors, sums) serve every shape."""
        trs = list(self.trainers.values())
        if not trs:
            return
        self.shared.build()                           # ONE potentials launch (and one expectations launch) for every shape
        together = [tr for tr in trs if tr.topo.P >= 1 and tr.batch.X == 64] if grouped_sweeps else []
        if len(together) < 2 and grouped_sweeps is not True:
            together = []
        if together:
            # shapes with ONE predicted word (no pairwise factor) join the call: the library hands all their graphs to the launches
            # it runs behind the matrix-core kernels anyway (the exact kernel over flagged graphs, their gradient) -- two launches
            # per such shape otherwise
            together = [tr for tr in trs if any(tr is t for t in together) or (tr.topo.P == 0 and tr.batch.X == 64)]
        for tr in trs:
            if not any(tr is t for t in together):
                tr._sweep_with_gradient()
        if together:
            from .batch import sweep_groups
            sweep_groups([tr.batch for tr in together], [tr.roots[:tr.n_sweeps_run] for tr in together], init=True,
                         marginals=[tr._marg for tr in together], gradients=[(tr._g_ee, tr._g_ed) for tr in together],
                         keep_messages=False)
        stats += self.shared.statistics(select)


class TiDirTrainer:
    """The outer loop of train_mp.py's __main__ (train_mp.py:560-690) over files in the reference's
    formats: vocabularies, feature matrices, JSON training instances -> one UserGraphTrainer per
    sentence shape sharing theta; one fused statistics buffer, one all-reduce and one update per epoch -- or, with
    `minibatch`, per minibatch of a shuffled epoch (the reference updates once per instance, train_mp.py:631-649 + 405-424:
    minibatch=1 in the limit); params written in the reference's text format, read back for a resumed run, predictions
    written in the reference's two text formats."""

    def __init__(self, ti_path, en_vocab, de_vocab, phi_pmi, phi_pmi_w1, phi_ed, phi_ped, device='cuda:0', sweeps=3,
                 rank=0, world=1, use_planes=True, adapt=None, domains=None, reg_param_ua_scale=1.0,
                 use_correct_feat=True, history=True, session_history=True, grouped_sweeps='auto', skip_unchanged=False,
                 minibatch=None, shuffle_seed=None, load_params=None, share_params_with=None, minibatch_mode='masked'):
        """use_planes switches the three per-instance feature planes on as a whole; use_correct_feat / history /
        session_history gate them one by one as the reference's options of the same names do (train_mp.py:178, 192,
        206: the 'correct', 'full_history' and 'hit_history' planes).
        adapt: None, 'user' (--user_adapt: domain = ti.user_id) or 'experience' (--experience_adapt: domain =
        len(ti.past_sentences_seen)), train_mp.py:162-171; `domains`: the domain names in file order (the
        reference reads <ti>.users / <ti>.experience, train_mp.py:504-516) -- default: the names that occur.
        minibatch: instances per update (None = the whole file: one update per epoch).  Every epoch walks the instances in
        a shuffled order (`random.shuffle(training_instances)`, train_mp.py:631) -- a permutation seeded by (shuffle_seed,
        epoch), the same on every rank; shuffle_seed=None keeps the file order -- and cuts it into minibatches; a minibatch is
        sharded contiguously over the ranks, its statistics all-reduced once, theta updated once.
        minibatch_mode: 'masked' (default) -- the rank's whole shard stays resident as ONE set of bucket trainers (one HIP graph);
        every minibatch evaluates ALL of them under the current thetas and sums the statistics of the minibatch's instances only
        (a device-side selection: the epoch's permutation is uploaded once, a minibatch costs one graph replay and no host work
        per instance; the surplus evaluations are what a GPU that is idle anyway pays for having nothing rebuilt).  'rebuild' --
        round 3's form: the bucket trainers of every minibatch are built from its instances (shape compile, table upload,
        programs), which costs tens of milliseconds of host time per minibatch but evaluates nothing twice: for shards far
        larger than a minibatch times the number of minibatches one can afford to replay.
        load_params: a params file (tidir.save_params / the reference's save_params) to start from instead of zeros
        (--load_params, train_mp.py:528-542: the adapt mode's extension is tried first, then the bare name).
        skip_unchanged: the sweeps drop updates that would recompute a message from unchanged inputs (include/mlbp.h
        MLBP_SWEEP_SKIP_UNCHANGED; equal to the full schedule to rounding, off by default like the C ABI).
        share_params_with: another TiDirTrainer whose theta tensors (global and per-domain) this one READS instead of owning
        its own -- the tune-set evaluator of train(tune=...), train_mp.py:657-684: same vocabularies, features and adapt mode."""
        from . import tidir
        self._files = dict(en_vocab=en_vocab, de_vocab=de_vocab, phi_pmi=phi_pmi, phi_pmi_w1=phi_pmi_w1, phi_ed=phi_ed, phi_ped=phi_ped)
        self._options = dict(sweeps=sweeps, use_planes=use_planes, reg_param_ua_scale=reg_param_ua_scale, use_correct_feat=use_correct_feat,
                             history=history, session_history=session_history, grouped_sweeps=grouped_sweeps, skip_unchanged=skip_unchanged)
        if adapt not in (None, 'user', 'experience'):
            raise ValueError("adapt is None, 'user' or 'experience'")
        self.adapt, self.reg_param_ua_scale = adapt, float(reg_param_ua_scale)
        self.grouped_sweeps, self.sweeps, self.skip_unchanged = grouped_sweeps, int(sweeps), bool(skip_unchanged)
        self.rank, self.world = int(rank), int(world)
        self.device = dev = torch.device(device)
        self.en, self.de = tidir.read_vocab(en_vocab), tidir.read_vocab(de_vocab)
        self.phi_ee, self.phi_w1, self.phi_ed_t = tidir.load_features(phi_pmi, phi_pmi_w1, phi_ed, phi_ped)
        self.instances = tidir.read_instances(ti_path)
        self.n_total = len(self.instances)
        self.use_planes = bool(use_planes)
        self.plane_features = {name: tidir.ED_NAMES.index(name)
                               for name, on in (('correct', use_correct_feat), ('full_history', history), ('hit_history', session_history)) if on}
        self.theta_en_en = torch.zeros(len(tidir.EE_NAMES), dtype=torch.float64, device=dev)   # train_mp.py:519-523
        self.theta_en_de = torch.zeros(len(tidir.ED_NAMES), dtype=torch.float64, device=dev)
        dom_of = (lambda r: str(r['user_id'])) if adapt == 'user' else (lambda r: str(r['n_seen']))
        self.domains = []
        if adapt:       # every rank needs the same list: it comes from ALL instances, not only this rank's shard
            if domains is None:
                en2id, de2id = {w: i for i, w in enumerate(self.en)}, {w: i for i, w in enumerate(self.de)}
                domains = sorted({dom_of(tidir.instance_shape(ti, en2id, de2id)[1]) for ti in self.instances})
            self.domains = [str(d) for d in domains]
            self.theta_dom_en_en = torch.zeros(len(self.domains), len(tidir.EE_NAMES), dtype=torch.float64, device=dev)  # train_mp.py:524-527
            self.theta_dom_en_de = torch.zeros(len(self.domains), len(tidir.ED_NAMES), dtype=torch.float64, device=dev)
        if share_params_with is not None:
            o = share_params_with
            if o.adapt != adapt or (adapt and list(o.domains) != list(self.domains)):
                raise ValueError('share_params_with: the other trainer has another adapt mode or domain list')
            self.theta_en_en, self.theta_en_de = o.theta_en_en, o.theta_en_de
            if adapt:
                self.theta_dom_en_en, self.theta_dom_en_de = o.theta_dom_en_en, o.theta_dom_en_de
        if load_params:
            self.load_params(load_params)
        self.minibatch = None if minibatch is None else max(1, int(minibatch))
        if minibatch_mode not in ('masked', 'rebuild'):
            raise ValueError("minibatch_mode is 'masked' or 'rebuild'")
        self.minibatch_mode = minibatch_mode
        self.shuffle_seed = shuffle_seed
        self.n_stat = len(tidir.EE_NAMES) + len(tidir.ED_NAMES) + 2
        self.stats = torch.zeros(self.n_stat * (1 + len(self.domains)), dtype=torch.float64, device=dev)
        self._epochs_done = 0
        # the whole shard as one set of buckets: full-batch epochs and the prediction pass
        lo, hi = mdist.shard_range(self.n_total, self.rank, self.world)
        self._shard = (lo, hi)
        self._full = _BucketSet(self, self.instances[lo:hi])
        self.buckets, self.trainers = self._full.buckets, self._full.trainers
        self._mini_sets = {}
        self._mgraph = None
        if self.minibatch is not None and self.minibatch_mode == 'masked':
            # device-side selection of a minibatch: which minibatch every instance of the file belongs to this epoch, the current
            # minibatch's number, and per bucket the file position of each of its instances
            self._mb_of = torch.zeros(max(self.n_total, 1), dtype=torch.int32, device=dev)
            self._m_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._acc = torch.zeros(2, dtype=torch.float64, device=dev)
            pos = [lo + r['index'] for key in self.trainers for r in self.buckets[key]['rows']]      # (bucket order = the set's instance order)
            self._file_pos = torch.from_numpy(np.array(pos, dtype=np.int64)).to(dev)
            self._key = torch.zeros(max(len(pos), 1), dtype=torch.int32, device=dev)      # minibatch of each resident instance this epoch

    # ---- parameters -----------------------------------------------------------------------------
    def load_params(self, path):
        """Starts from a params file (train_mp.py:528-542): '<path><ext>' with the adapt mode's extension first, then
        '<path>'.  Global thetas, and every adapted domain's thetas the file holds for a domain this run knows."""
        from . import tidir
        ext = {'user': '.user_adapt', 'experience': '.exp_adapt', None: ''}[self.adapt]
        import os
        chosen = path + ext if os.path.exists(path + ext) else path
        een, eet, edn, edt, d2t = tidir.read_params(chosen)
        if list(een) != list(tidir.EE_NAMES) or list(edn) != list(tidir.ED_NAMES):
            raise ValueError('params file %s names other features than this build' % chosen)
        self.theta_en_en.copy_(torch.from_numpy(np.asarray(eet, dtype=np.float64).reshape(-1)))
        self.theta_en_de.copy_(torch.from_numpy(np.asarray(edt, dtype=np.float64).reshape(-1)))
        for i, d in enumerate(self.domains):
            if ('en_en', d) in d2t:
                self.theta_dom_en_en[i].copy_(torch.from_numpy(np.asarray(d2t['en_en', d], dtype=np.float64).reshape(-1)))
            if ('en_de', d) in d2t:
                self.theta_dom_en_de[i].copy_(torch.from_numpy(np.asarray(d2t['en_de', d], dtype=np.float64).reshape(-1)))
        return chosen

    def domain_thetas(self):
        """{('en_en' | 'en_de', domain name): (1, F) array} as train_mp's domain2theta / the params file."""
        d2t = {}
        for i, d in enumerate(self.domains):
            d2t['en_en', d] = self.theta_dom_en_en[i].cpu().numpy().reshape(1, -1)
            d2t['en_de', d] = self.theta_dom_en_de[i].cpu().numpy().reshape(1, -1)
        return d2t

    # ---- statistics -----------------------------------------------------------------------------
    def capture(self):
        """Records local_statistics() -- every bucket's potentials, sweeps (grouped or not), gradients and sums -- into one
        HIP graph; later calls replay it.  theta is read from the same device tensors at every replay.  One eager pass
        first: first-use allocations, attribute calls and the group table's upload must not be captured."""
        self._graph = None
        self._local_statistics_eager()
        torch.cuda.synchronize(self.theta_en_en.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._local_statistics_eager()
        self._graph = g
        return self

    def local_statistics(self):
        """Sum of this rank's buckets' statistics over its whole shard."""
        if getattr(self, '_graph', None) is not None:
            self._graph.replay()
            return self.stats
        return self._local_statistics_eager()

    def _local_statistics_eager(self, bucket_set=None):
        self.stats.zero_()
        (bucket_set or self._full).statistics_into(self.stats, self.grouped_sweeps)
        return self.stats

    def _update(self, learning_rate, reg_param):
        """One all-reduce of the fused buffer (global and per-domain statistics), one update of every theta."""
        mdist.all_reduce_sum_(self.stats)
        n, F_ee, F_ed = self.n_stat, len(self.theta_en_en), len(self.theta_en_de)
        apply_update(self.theta_en_en, self.theta_en_de, self.stats[:n], F_ee, F_ed, learning_rate, reg_param)
        if self.domains:
            apply_domain_update(self.theta_dom_en_en, self.theta_dom_en_de, self.stats[n:].view(len(self.domains), n), F_ee, F_ed,
                                learning_rate, reg_param * self.reg_param_ua_scale)
        return float(self.stats[n - 2].item()), float(self.stats[n - 1].item())

    def epoch_order(self, epoch):
        """The order in which epoch `epoch` walks ALL instances (every rank computes the same one)."""
        if self.shuffle_seed is None:
            return np.arange(self.n_total)
        return np.random.RandomState([int(self.shuffle_seed) & 0x7FFFFFFF, int(epoch)]).permutation(self.n_total)

    def _minibatch_set(self, epoch, m, ids):
        """The bucket trainers of this rank's share of one minibatch (built once per (order, minibatch): a fixed order reuses
        them every epoch)."""
        key = (epoch if self.shuffle_seed is not None else 0, m)
        if key not in self._mini_sets:
            if self.shuffle_seed is not None:
                self._mini_sets = {k: v for k, v in self._mini_sets.items() if k[0] == key[0]}      # last epoch's sets go
            lo, hi = mdist.shard_range(len(ids), self.rank, self.world)
            self._mini_sets[key] = _BucketSet(self, [self.instances[int(i)] for i in ids[lo:hi]])
        return self._mini_sets[key]

    def _masked_statistics(self):
        """Statistics of the current minibatch (self._m_dev) over this rank's resident shard."""
        if self._mgraph is not None:
            self._mgraph.replay()
            return self.stats
        self.stats.zero_()
        self._full.statistics_into(self.stats, self.grouped_sweeps, select=(self._key, self._m_dev))
        return self.stats

    def capture_masked(self):
        """Records _masked_statistics() into one HIP graph: the minibatch number and the thetas are read from device tensors
        at every replay, so ONE graph serves every minibatch of every epoch."""
        self._mgraph = None
        self._masked_statistics()
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._masked_statistics()
        self._mgraph = g
        return self

    def _epoch_masked(self, epoch, learning_rate, reg_param):
        order = self.epoch_order(epoch)
        mb = np.empty(self.n_total, dtype=np.int32)
        mb[order] = np.arange(self.n_total, dtype=np.int64) // self.minibatch
        self._mb_of.copy_(torch.from_numpy(mb))                  # the epoch's permutation: one upload
        torch.index_select(self._mb_of, 0, self._file_pos, out=self._key)
        self._acc.zero_()
        for m in range((self.n_total + self.minibatch - 1) // self.minibatch):
            self._m_dev.fill_(m)
            self._masked_statistics()
            self._update_on_device(learning_rate, reg_param)
        lp, cnt = (float(v) for v in self._acc.cpu())                # (read once per epoch: no host round trip per minibatch)
        return lp / max(cnt, 1.0)

    def _update_on_device(self, learning_rate, reg_param):
        """_update() without its host read-back: all-reduce, theta updates, the running [sum log-posterior, count]."""
        mdist.all_reduce_sum_(self.stats)
        n, F_ee, F_ed = self.n_stat, len(self.theta_en_en), len(self.theta_en_de)
        apply_update(self.theta_en_en, self.theta_en_de, self.stats[:n], F_ee, F_ed, learning_rate, reg_param)
        if self.domains:
            apply_domain_update(self.theta_dom_en_en, self.theta_dom_en_de, self.stats[n:].view(len(self.domains), n), F_ee, F_ed,
                                learning_rate, reg_param * self.reg_param_ua_scale)
        self._acc += self.stats[n - 2:n]

    def epoch(self, learning_rate, reg_param):
        """One pass over the instances; returns the mean log-posterior (at the thetas each instance was evaluated under)."""
        epoch = self._epochs_done
        self._epochs_done += 1
        if self.minibatch is None:
            self.local_statistics()
            lp, n = self._update(learning_rate, reg_param)
            return lp / max(n, 1.0)
        if self.minibatch_mode == 'masked':
            return self._epoch_masked(epoch, learning_rate, reg_param)
        order = self.epoch_order(epoch)
        lp_sum, n_sum = 0.0, 0.0
        for m, m0 in enumerate(range(0, self.n_total, self.minibatch)):
            self._local_statistics_eager(self._minibatch_set(epoch, m, order[m0:m0 + self.minibatch]))
            lp, n = self._update(learning_rate, reg_param)
            lp_sum += lp; n_sum += n
        return lp_sum / max(n_sum, 1.0)

    def tune_evaluator(self, tune_path):
        """A second trainer over the instances of `tune_path` (--tune, train_mp.py:463, 571-580) that reads THIS trainer's thetas:
        its predict() is the per-epoch tune evaluation of train_mp.py:657-684."""
        return TiDirTrainer(tune_path, self._files['en_vocab'], self._files['de_vocab'], self._files['phi_pmi'], self._files['phi_pmi_w1'],
                            self._files['phi_ed'], self._files['phi_ped'], device=self.device, rank=self.rank, world=self.world,
                            adapt=self.adapt, domains=self.domains if self.adapt else None, share_params_with=self, **self._options)

    def train(self, epochs=3, reg_param=0.2, save_params=None, capture=True, tune=None):
        """lr = 0.1 / (1 + 0.3 epoch) (train_mp.py:627-630); regularisation reg_param / N (train_mp.py:160);
        params saved as <save_params><ext>.iter<epoch> and <save_params><ext> with ext = '.user_adapt' / '.exp_adapt' /
        '' by adapt mode (train_mp.py:654-656, 688-690).
        tune: a TI file (or a tune_evaluator()) evaluated after EVERY epoch under the thetas that epoch left (train_mp.py:657-684:
        `batch_predictions` over --tune, then the mean log-posterior and precision at 0 / 25 / 50); the per-epoch results --
        (mean log-posterior, (p@0, p@25, p@50, total)) -- are kept in self.tune_history.
        capture: whole-file epochs (minibatch=None) replay one HIP graph of the step's launches (capture(): same bits as the
        eager launches, 7 % less time per step at 8192 instances) when there are at least three of them."""
        from . import tidir
        if capture and self.minibatch is None and epochs >= 3 and getattr(self, '_graph', None) is None and self.trainers:
            self.capture()
        if capture and self.minibatch is not None and self.minibatch_mode == 'masked' and self._mgraph is None and self.trainers and \
                epochs * ((self.n_total + self.minibatch - 1) // self.minibatch) >= 3:
            self.capture_masked()
        if save_params:
            save_params = save_params + {'user': '.user_adapt', 'experience': '.exp_adapt', None: ''}[self.adapt]
        history = []
        tuner = self.tune_evaluator(tune) if isinstance(tune, str) else tune
        self.tune_history = []
        for epoch in range(epochs):
            history.append(self.epoch(0.1 / (1.0 + 0.3 * epoch), float(reg_param) / float(self.n_total)))
            if save_params and self.rank == 0:
                tidir.save_params('%s.iter%d' % (save_params, epoch), self.theta_en_en.cpu().numpy().reshape(1, -1),
                                  self.theta_en_de.cpu().numpy().reshape(1, -1), d2t=self.domain_thetas())
            if tuner is not None:
                self.tune_history.append(tuner.predict())
        if save_params and self.rank == 0:
            tidir.save_params(save_params, self.theta_en_en.cpu().numpy().reshape(1, -1),
                              self.theta_en_de.cpu().numpy().reshape(1, -1), d2t=self.domain_thetas())
        return history

    # ---- prediction pass (train_mp.py:310-343, 692-770) -------------------------------------------
    def predict(self, save_predictions=None):
        """-> (mean log-posterior, (p@0, p@25, p@50, total)) over ALL ranks' instances (train_mp.py:666-684: sums reduced once).
        save_predictions: the '*SENT_ID:' blocks go to '<path><ext>' and the '.dist' lines to '<path><ext>.dist', one block per
        instance in file order, exactly the text `batch_predictions` returns (train_mp.py:337-338, 752-757) -- formed from the
        batched top-50 indices and log-marginals, no per-instance graph.  With several ranks every rank writes its shard
        ('.rank<r>') and rank 0 joins the shards into the ONE pair of files the reference writes (train_mp.py:740-760)."""
        from . import tidir
        lp, counts = 0.0, np.zeros(4, dtype=np.int64)
        blocks = {}
        for key, tr in self.trainers.items():
            l, idx, logs, c, logm, label_logs = tr.predict(top=min(50, tr.batch.X), with_logs=True)
            lp += float(l.sum()); counts += np.array(c)
            if save_predictions:
                rows = self.buckets[key]['rows']
                for b, row in enumerate(rows):
                    blocks[row['index']] = tidir.prediction_text(row, list(key[1]), self.en, idx[b], logs[b], label_logs[b], logm[b])
        # the mean's denominator is every instance of the shard -- `/ float(len(testing_instances))`, train_mp.py:684, 764 -- also
        # the ones without a predicted word, which build no graph here (the reference's workers fail on them and the pool drops
        # the task silently, train_mp.py:666-680; LBP.py:193-194)
        n = self._shard[1] - self._shard[0]
        if save_predictions:
            import codecs
            ext = {'user': '.user_adapt', 'experience': '.exp_adapt', None: ''}[self.adapt]
            tail = '.rank%d' % self.rank if self.world > 1 else ''
            with codecs.open(save_predictions + ext + tail, 'w', 'utf8') as w, codecs.open(save_predictions + ext + '.dist' + tail, 'w', 'utf8') as wd:
                for i in sorted(blocks):
                    w.write(blocks[i][0] + '\n')
                    wd.write(blocks[i][1] + '\n')
        tot = torch.tensor([lp, float(n)] + [float(v) for v in counts], dtype=torch.float64, device=self.device)
        mdist.all_reduce_sum_(tot)
        tot = tot.cpu().numpy()
        if save_predictions and self.world > 1:
            # ONE '<path><ext>' and one '.dist' in instance order, as the reference writes them (train_mp.py:740-760) and eval.py /
            # get_acc.py read them: the shards are contiguous, so rank order is instance order.  (The all-reduce above is the
            # barrier: every rank has closed its files.)
            import shutil
            ext = {'user': '.user_adapt', 'experience': '.exp_adapt', None: ''}[self.adapt]
            if self.rank == 0:
                for suffix in ('', '.dist'):
                    with open(save_predictions + ext + suffix, 'wb') as out:
                        for r in range(self.world):
                            part = '%s%s%s.rank%d' % (save_predictions, ext, suffix, r)
                            with open(part, 'rb') as f:
                                shutil.copyfileobj(f, out)
                            os.remove(part)
            mdist.barrier()
        return float(tot[0] / max(tot[1], 1.0)), tuple(int(v) for v in tot[2:])
