"""Diagnostic: 30 eager train-shaped steps (UserGraphTrainer.local_statistics, shared pots, B = 8192) for
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/prof_train_step.py` -- the per-kernel split
of one optimisation step quoted in DESIGN.md."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import cases as C
from macaronicusermodeling_amd.train import UserGraphTrainer
B, X = 8192, 64
spec = C.user_spec(10, [1, 3, 5, 8], X, 64, seed=2) if '--k4' in sys.argv else C.user_spec(10, [1, 4, 7], X, 64, seed=1)
from macaronicusermodeling_amd.topology import GraphTopology
topo = GraphTopology.from_spec(spec)
inputs = C.make_inputs(spec, 5)
if '--random-planes' not in sys.argv:        # the reference's tensors: [pmi, 0, 1] and [pmi, pmi_w1, 1] (train_mp.py:600-606)
    inputs = C.reference_planes(inputs)
rs = np.random.RandomState(0)
labels = rs.randint(0, X, size=(B, topo.n_vars)); obs = rs.randint(0, 64, size=(B, topo.U))
tr = UserGraphTrainer(spec, labels, obs, inputs['phi_en_en'], inputs['phi_en_en_w1'], inputs['phi_en_de'],
                      inputs['theta_en_en'], inputs['theta_en_de'])
for _ in range(30):
    tr.local_statistics()
torch.cuda.synchronize()
