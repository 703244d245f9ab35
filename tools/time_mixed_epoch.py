"""Diagnostic: TiDirTrainer.epoch on a synthetic TI_DIR whose sentences have 2-4 predicted words (K2 / K3 / K4 buckets in one
launch sequence: mlbp_sweep_groups_f64 runs one sweep launch per form of the shared-table kernel).  With
MLBP_SHARED_NO_PF=1 MLBP_SHARED_NO_P3=1 in the environment every group takes the general form (what a mixed launch ran before)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from macaronicusermodeling_amd import tidir
from macaronicusermodeling_amd.train import TiDirTrainer
dev = torch.device('cuda:0')
d = tempfile.mkdtemp()
lo = 1 if '--k1' in sys.argv else 2          # --k1: sentences with a single predicted word (no pairwise factor) among them
paths = tidir.synthesize(d, n_instances=8192, X=64, Vde=64, sent_len=(6, 9), n_predicted=(lo, 4), seed=21)
tt = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'],
                  device=dev, sweeps=3)
lr, reg = 1e-6, 0.2 / tt.n_total
shapes = {}
for tr in tt.trainers.values():
    shapes[tr.topo.P] = shapes.get(tr.topo.P, 0) + tr.batch.B
print('instances', tt.n_total, 'shapes', len(tt.trainers), 'instances by pairwise factors', dict(sorted(shapes.items())))
def timed(n=5):
    tt.epoch(lr, reg); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n): tt.epoch(lr, reg)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3
print('whole-file epoch, eager: %.3f ms' % timed())
tt.capture()
print('whole-file epoch, one HIP graph replay: %.3f ms' % timed(20))
