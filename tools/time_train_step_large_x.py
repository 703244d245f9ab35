#!/usr/bin/env python3
"""Times the train-shaped step at a large state space (X = 512, shared pots): sweeps as batched DGEMMs, pairwise part
of the gradient as DGEMMs, against the same step on the per-graph kernels (variant 3 + per-graph gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import cases as C
from macaronicusermodeling_amd import _ffi
from macaronicusermodeling_amd.train import UserGraphTrainer
from macaronicusermodeling_amd.topology import GraphTopology

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

X, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = C.user_spec(10, [1, 4, 7], X, 64, seed=1)
topo = GraphTopology.from_spec(spec)
inputs = C.make_inputs(spec, 5)
rs = np.random.RandomState(0)
labels = rs.randint(0, X, size=(B, topo.n_vars))
by_id = {f['id']: f for f in spec['factors']}
obs = np.stack([rs.randint(0, 64 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else X, size=B) for j in topo.unary_factors], axis=1)
tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                      inputs['theta_en_en'] * 0.05, inputs['theta_en_de'] * 0.05)
t_fast = timed(tr.local_statistics)
fast = tr.local_statistics().clone()
print('X=%d B=%d shared pots: local_statistics %.2f ms (sweeps kernel family %d)' % (X, B, t_fast, _ffi.lib.mlbp_last_sweep_kernel()))
_ffi.check(_ffi.lib.mlbp_set_sweep_variant(3)); tr.batch.use_shared_gradient = False
t_slow = timed(tr.local_statistics, n=1)
slow = tr.local_statistics().clone()
print('same step on the per-graph kernels: %.2f ms; max relative difference of the statistics %.2e' % (
    t_slow, float(((fast - slow).abs() / slow.abs().clamp_min(1e-30)).max())))
