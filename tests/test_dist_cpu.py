"""N > 1 path on CPU: world_size-2 gloo process group.  Graphs shard contiguously with no data-path
collective; the only exchange is ONE all-reduce of the fused statistics buffer per step
(macaronicusermodeling_amd.dist / train.apply_update).  Per-rank statistics come from the oracle
here (no GPU in this suite); the GPU twin is tests/test_gpu_gradient.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import cases as C
from oracle import lbp_oracle as O


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _stats_for(spec, inputs, seeds, roots):
    """[grad_en_en | grad_en_de | log-posterior | count] summed over instances (unregularised)."""
    g = O.Graph(spec)
    out = np.zeros(C.F_EE + C.F_ED + 2)
    for sd in seeds:
        inp = dict(inputs)
        rs = np.random.RandomState(sd)
        inp['pot_en_en'] = inputs['pot_en_en'] * (1 + 0.1 * rs.rand(*inputs['pot_en_en'].shape))
        msgs = O.init_messages(g)
        O.treelike_inference(g, inp, msgs, len(roots), roots, True)
        ee, ed = O.unregularized_gradient(g, inp, msgs)
        out[:C.F_EE] += ee.reshape(-1); out[C.F_EE:C.F_EE + C.F_ED] += ed.reshape(-1)
        out[-2] += O.log_posterior(g, msgs); out[-1] += 1
    return out


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from macaronicusermodeling_amd import dist as mdist
    from macaronicusermodeling_amd.train import apply_update
    r, w, _ = mdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and mdist.world_size() == world
    spec = C.user_spec(6, [0, 2, 3], 8, 8, seed=3)
    inputs = C.make_inputs(spec, 11)
    lo, hi = mdist.shard_range(n_items, rank, world)
    # the trainer's fused buffer (train.UserGraphTrainer.stats_all): [global statistics | per-domain statistics [D][n_stat]],
    # ONE all-reduce for both halves; item i belongs to domain i % D
    n_stat, D = C.F_EE + C.F_ED + 2, 3
    fused = torch.zeros(n_stat * (1 + D), dtype=torch.float64)
    for i in range(lo, hi):
        s_i = torch.from_numpy(_stats_for(spec, inputs, [100 + i], [0, 2]))
        fused[:n_stat] += s_i
        fused[n_stat * (1 + i % D):n_stat * (2 + i % D)] += s_i
    mdist.all_reduce_sum_(fused)
    stats = fused[:n_stat]
    t_ee = torch.from_numpy(inputs['theta_en_en'].reshape(-1).copy())
    t_ed = torch.from_numpy(inputs['theta_en_de'].reshape(-1).copy())
    apply_update(t_ee, t_ed, stats, C.F_EE, C.F_ED, 0.1, 0.01)
    from macaronicusermodeling_amd.train import apply_domain_update
    d_ee, d_ed = torch.zeros(D, C.F_EE, dtype=torch.float64) + 0.25, torch.zeros(D, C.F_ED, dtype=torch.float64) - 0.5
    apply_domain_update(d_ee, d_ed, fused[n_stat:].view(D, n_stat), C.F_EE, C.F_ED, 0.1, 0.01 * 0.5)
    q.put((rank, lo, hi, stats.numpy().copy(), t_ee.numpy().copy(), t_ed.numpy().copy(), d_ee.numpy().copy(), d_ed.numpy().copy()))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_step_equals_single_process():
    world, n_items = 2, 7
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]          # contiguous shards, remainder to rank 0
    spec = C.user_spec(6, [0, 2, 3], 8, 8, seed=3)
    inputs = C.make_inputs(spec, 11)
    want = _stats_for(spec, inputs, range(100, 100 + n_items), [0, 2])
    for r in res:
        np.testing.assert_allclose(r[3], want, rtol=1e-12)
        np.testing.assert_array_equal(r[4], res[0][4])               # every rank ends with the same theta
    n = want[-1]
    np.testing.assert_allclose(res[0][4], inputs['theta_en_en'].reshape(-1) +
                               0.1 * (want[:C.F_EE] - n * 0.01 * inputs['theta_en_en'].reshape(-1)), rtol=1e-12)
    # the per-domain half of the same buffer: domain d = the items i with i % 3 == d, wherever they were sharded
    for d in range(3):
        w_d = _stats_for(spec, inputs, [100 + i for i in range(n_items) if i % 3 == d], [0, 2])
        for r in res:
            np.testing.assert_allclose(r[6][d], 0.25 + 0.1 * (w_d[:C.F_EE] - w_d[-1] * 0.005 * 0.25), rtol=1e-12)
            np.testing.assert_allclose(r[7][d], -0.5 + 0.1 * (w_d[C.F_EE:C.F_EE + C.F_ED] - w_d[-1] * 0.005 * -0.5), rtol=1e-12)


def test_shard_range_covers_everything_once():
    from macaronicusermodeling_amd.dist import shard_range
    for n in (0, 1, 7, 8, 8192, 8195):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_domain_update_rule():
    """apply_domain_update == the per-domain half of batch_sgd_accumulate (train_mp.py:384-396, 413-415):
    theta_d += sum_{i in d} lr (g_i - reg theta_d), written as lr (sum g - n_d reg theta_d); a domain with no
    instance on any rank keeps its theta."""
    from macaronicusermodeling_amd.train import apply_domain_update
    rs = np.random.RandomState(0)
    D, lr, reg = 4, 0.07, 0.03
    th_ee, th_ed = rs.randn(D, C.F_EE), rs.randn(D, C.F_ED)
    counts = [3, 0, 1, 5]
    g = [[rs.randn(C.F_EE + C.F_ED) for _ in range(n)] for n in counts]
    stats = np.zeros((D, C.F_EE + C.F_ED + 2))
    want_ee, want_ed = th_ee.copy(), th_ed.copy()
    for d in range(D):
        for gi in g[d]:
            stats[d, :C.F_EE + C.F_ED] += gi; stats[d, -1] += 1
            want_ee[d] += lr * (gi[:C.F_EE] - reg * th_ee[d])        # apply_regularization per instance, same theta
            want_ed[d] += lr * (gi[C.F_EE:] - reg * th_ed[d])
    t_ee, t_ed = torch.from_numpy(th_ee.copy()), torch.from_numpy(th_ed.copy())
    apply_domain_update(t_ee, t_ed, torch.from_numpy(stats), C.F_EE, C.F_ED, lr, reg)
    np.testing.assert_allclose(t_ee.numpy(), want_ee, rtol=1e-12)
    np.testing.assert_allclose(t_ed.numpy(), want_ed, rtol=1e-12)
    np.testing.assert_array_equal(t_ee.numpy()[1], th_ee[1])
