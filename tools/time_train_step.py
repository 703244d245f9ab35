#!/usr/bin/env python3
"""Times the train_mp-shaped step (potentials -> sweeps -> gradient -> reduce) per kernel with HIP events.
Two table modes: 'shared' = the real train_mp layout (two shared pots + transposed columns), 'unique' = one table
per (graph, factor) as in bench.py (FactorGraphBatch driven directly)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import cases as C
from macaronicusermodeling_amd.train import UserGraphTrainer
from macaronicusermodeling_amd.batch import FactorGraphBatch
from macaronicusermodeling_amd.topology import GraphTopology

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

B, X = 8192, 64
# --k4: four predicted words (six pairwise factors, the shared-table kernel's general instance: spilled tiles, three-source updates)
# --k4-adjacent / --k3-adjacent: neighbouring predicted words, so that both en_en pots (gap 1 / gap > 1) are in use and a bundle's two
# updates may need fragment sets that one half of the workgroup holds
if '--k1' in sys.argv: spec = C.user_spec(10, [4], X, 64, seed=2)
elif '--k2' in sys.argv: spec = C.user_spec(10, [2, 6], X, 64, seed=2)
elif '--k5' in sys.argv: spec = C.user_spec(10, [1, 3, 5, 7, 9], X, 64, seed=2)
elif '--k6' in sys.argv: spec = C.user_spec(10, [0, 2, 4, 5, 7, 9], X, 64, seed=2)
elif '--k4-adjacent' in sys.argv: spec = C.user_spec(10, [1, 2, 5, 6], X, 64, seed=2)
elif '--k3-adjacent' in sys.argv: spec = C.user_spec(10, [1, 2, 7], X, 64, seed=1)
elif '--k4' in sys.argv: spec = C.user_spec(10, [1, 3, 5, 8], X, 64, seed=2)
else: spec = C.user_spec(10, [1, 4, 7], X, 64, seed=1)
topo = GraphTopology.from_spec(spec)
inputs = C.make_inputs(spec, 5)
if '--random-planes' not in sys.argv:        # the reference's tensors: [pmi, 0, 1] and [pmi, pmi_w1, 1] (train_mp.py:600-606)
    inputs = C.reference_planes(inputs)
rs = np.random.RandomState(0)
labels = rs.randint(0, X, size=(B, topo.n_vars)); obs = rs.randint(0, 64, size=(B, topo.U))
tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                      inputs['theta_en_en'], inputs['theta_en_de'])
for skip in (False, True):
    tr.batch.skip_unchanged = skip
    print('shared pots:  full local_statistics %.3f ms (skip_unchanged=%s)' % (timed(tr.local_statistics), skip))
eager = tr.local_statistics().clone()
if '--short' in sys.argv: tr.capture(); print('shared pots:  full local_statistics as one HIP graph replay %.3f ms' % timed(tr.local_statistics)); sys.exit(0)
tr.capture()
print('shared pots:  full local_statistics as one HIP graph replay %.3f ms' % timed(tr.local_statistics))
assert torch.equal(eager, tr.local_statistics()), 'graph replay differs from eager'
tr._graph = None
print('   potentials %.3f  sweep+marginals %.3f  gradient %.3f' % (
    timed(tr.build_potentials), timed(lambda: tr.batch.sweep(tr.roots, init=True, marginals=tr._marg)),
    timed(lambda: tr.batch.gradient(tr._g_ee, tr._g_ed))))
if '--k4' in sys.argv:
    sys.exit(0)
# unique tables
dev = torch.device('cuda:0')
fb = FactorGraphBatch(topo, X, B, device=dev)
fb.set_pair_tables(torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev) + 0.01)
fb.set_unary_tables(torch.rand(B * topo.U, X, dtype=torch.float64, device=dev) + 0.01)
by_id = {f['id']: f for f in spec['factors']}
pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
kinds = [2 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else (0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1) for j in topo.unary_factors]
fb.set_features(inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'], pair_phi, kinds)
fb.set_observations(labels, obs)
marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
gee = torch.empty(B, 3, dtype=torch.float64, device=dev); ged = torch.empty(B, 6, dtype=torch.float64, device=dev)
for planar in (True, False, True, False):
  fb.use_planar = planar
  print('planar' if planar else 'interleaved', end=' ')
  print('unique tables: sweep+marginals %.3f ms   gradient alone %.3f ms   sweep+marginals+fused gradient %.3f ms (skip_unchanged: %.3f ms)' % (
      timed(lambda: fb.sweep([1, 4, 7], init=True, marginals=marg)), timed(lambda: fb.gradient()),
      timed(lambda: fb.sweep([1, 4, 7], init=True, marginals=marg, gradient=(gee, ged))),
      timed(lambda: fb.sweep([1, 4, 7], init=True, marginals=marg, gradient=(gee, ged), skip_unchanged=True))))
