#!/usr/bin/env python3
"""Benchmark of the batched LBP sweep (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          (WORLD_SIZE unset: starts its own N rank processes, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default): BASELINE.json configs[2] -- the configuration the metric string is quoted on
("batch=8192 graphs |X|=64"): 8192 train_mp-shaped user graphs PER GPU (sentence of 10 words, 3
predicted + 7 given => K3 of pairwise factors + 24 unary factors, train_mp.py:257-299), |X|=64,
float64, a UNIQUE pairwise table per (graph, factor) so the update kernel is HBM-bound
(SURVEY.md section 8(d)).  One *step* is what the reference does per instance between building the
graph and reading the posterior: `initialize` + `treelike_inference(3)` (train_mp.py:381-382) +
`get_posterior_probs` (train_mp.py:400), over the whole batch; with N > 1 the per-step statistics
vector is all-reduced over RCCL like train_mp.py's accumulate callback (train_mp.py:405-424).
One *iter* of the metric is one sweep (up + down pass, LBP.py:227-243) over every graph of the
batch, so a step contributes `sweeps` iters.

value = whole-job iters/s x (graphs per GPU / 8192) aggregated over GPUs -- i.e. sweeps of an
8192-graph batch per second; with N GPUs the job holds N batches (weak scaling).

Extra objects on the JSON line: `roofline` (dominant kernel = the fused sweep launch) and `cpu_baseline`
(the CPU oracle -- a port of the reference's per-graph, per-message NumPy cost model -- timed on this
box's host cores on a bounded sample; rank 0, N=1 only).

roofline, HBM-bound workloads: `achieved` = COMPULSORY HBM bytes of one launch / HIP-event time of that
launch, where compulsory = every distinct table and unary row the batch references read once + the index
arrays + the message write-back + the marginals (`compulsory_bytes` breaks it down so that it can be
recomputed by hand); `frac` = achieved / 8 TB/s and is <= 1 by construction.  The resident-table kernels
read each table once per launch however many sweeps reuse it, so SURVEY.md 8(d)'s per-update "algorithmic"
figure (a table counted once per UPDATE) exceeds the bytes that move; it is reported beside the fraction as
`algorithmic_GBps` with `table_reuse_factor` = algorithmic / compulsory bytes.  `traffic` = HBM bytes per
launch measured with rocprofv3 --pmc (tools/pmc_passes.sh -> profiles/pmc_traffic.json); it is reported only
when that measurement was made on the kernel sources of this run (sha of csrc/), else null with the reason.

Before the W warmup steps the same step runs 300 more times untimed (MLBP_BENCH_SPINUP_STEPS) so that the GPU, idle
since process start, is at its steady clocks when the timed region begins.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

KERNEL_NAMES = {2: 'sweep_x64_fused_kernel (exact)',
                3: 'shared_prepare_kernel + sweep_x64_shared_kernel (v_mfma_f64_16x16x4_f64; product-fused form where every variable has at most two pairwise factors, its three-source variant for K4 cliques) + the fix-up pass: the timed region is the whole launch sequence',
                4: 'sweep_wide_kernel', 5: 'sweep_generic_kernel',
                7: 'sweep_x64_lean_kernel (scale-free, micro-op form) + the ~5 us fix-up pass of sweep_x64_fused_kernel, timed together',
                6: 'contract_kernel (mlbp_gemm.hip): one hand-written MFMA launch per factor->variable update over the whole batch, variable product and renormalisation fused (f64: v_mfma_f64_16x16x4_f64; f32 tables: v_mfma_f32_16x16x4_f32)'}
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32-input MFMA = the f32 vector rate (155 TF measured)
F64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix spec (= FP64 vector; 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz); the guide's
                                # MFMA table has no f64 row


def workload_spec(name):
    import cases as C
    if name in ('user_k3', 'user_k3_shared', 'user_k3_trainlayout'):
        return C.user_spec(10, [1, 4, 7], 64, 64, seed=1), [1, 4, 7], 3, 1236
    if name in ('user_k4', 'user_k4_shared', 'user_k4_trainlayout'):     # four predicted words: K4 = 6 pairwise + 28 unary factors
        return C.user_spec(10, [1, 3, 5, 8], 64, 64, seed=2), [1, 3, 5], 3, 1237
    if name == 'chain8':
        return C.chain_spec(8, 64), [0] * 10, 10, 1235
    if name == 'ring8':
        return C.ring_spec(8, 64), [0] * 10, 10, 1235
    if name in ('ring8_x1000', 'ring8_x1000_shared'):      # a vocabulary-sized state space that is not a power of two
        return C.ring_spec(8, 1000), [0] * 10, 10, 1239
    if name == 'ring8_x2048_shared':    # past the one-image contraction kernel: 2 x 2 blocks of 1024 states
        return C.ring_spec(8, 2048), [0] * 10, 10, 1240
    if name in ('ring8_x512', 'ring8_x512_f32', 'ring8_x512_shared', 'ring8_x512_shared_f32'):
        return C.ring_spec(8, 512), [0] * 10, 10, 1238
    raise SystemExit('unknown workload %s' % name)


def algorithmic_bytes_per_graph(topo, roots, X, elem=8, table_elem=8):
    """pairwise update (X^2 + 2X) s; variable update (d+1) X s; unary 2X s.  In the
    float32-table mode the table term counts 4-byte entries, the messages stay float64."""
    from macaronicusermodeling_amd import _ffi
    total = 0
    for r in roots:
        ops, _ = topo.compile_sweep(r)
        for kind, a, b, c in ops.tolist():
            if kind in (_ffi.OP_PAIR_TM, _ffi.OP_PAIR_MT):
                total += X * X * table_elem + 2 * X * elem
            elif kind == _ffi.OP_VAR:
                total += (b + 1) * X * elem
            else:
                total += 2 * X * elem
    return total


def compulsory_bytes(topo, X, B, n_pair_tables_used, n_unary_rows_used, table_elem, keep_messages, init, marginals):
    """HBM bytes one sweep launch cannot avoid: every distinct table / unary row the batch references once, the
    int32 index arrays, the messages (read unless the launch initialises them, written back unless told not to)
    and the marginal read-out."""
    parts = {
        'pair_tables': n_pair_tables_used * X * X * table_elem,
        'unary_rows': n_unary_rows_used * X * 8,
        'index_arrays': B * (topo.P + topo.U) * 4,
        'message_read': 0 if init else B * topo.n_msgs * X * 8,
        'message_writeback': B * topo.n_msgs * X * 8 if keep_messages else 0,
        'marginals': B * topo.n_vars * X * 8 if marginals else 0,
    }
    parts['total'] = sum(parts.values())
    return parts


def kernel_sources_sha():
    """sha256 over the HIP/C++ sources the library is built from: ties a PMC traffic measurement to the kernels it
    was made on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'macaronicusermodeling_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.cpp', '.h')):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def self_launch(a, argv):
    """`python bench.py --gpus N` with no launcher around it: this process (which has not imported torch and never
    touches the GPU) starts N fresh rank processes through torch.distributed.run -- one per GPU, RCCL rendezvous on
    127.0.0.1 -- waits for them and relays rank 0's JSON line.  Nothing is re-exec'ed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)


# ---------------------------------------------------------------------------------------------------
# The outer loop's workload (secondary figures, never `value`): a synthetic TI_DIR of 8192 instances, sentences of eight words
# with three predicted ones at any positions = 56 sentence shapes (all K3 over the two shared pots), train_mp.py:560-690's epoch.
# ---------------------------------------------------------------------------------------------------
TRAIN_EPOCH_TIDIR = dict(n_instances=8192, X=64, Vde=64, sent_len=(8, 8), n_predicted=(3, 3), seed=11)


def _cpu_epoch_worker(args):
    """The reference's per-instance cost model (batch_sgd, train_mp.py:362-402) by the oracle: graph of the instance,
    initialize, treelike_inference(3), return_gradient, get_posterior_probs -- one instance after the other."""
    specs, rows, inputs = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    import copy
    from oracle import lbp_oracle as O
    n = 0
    for spec, labels, obs in rows:
        s = copy.deepcopy(specs[spec])
        g0 = O.Graph(s)
        s['labels'] = [int(labels[g0.var_order.index(v)]) for v in s['var_ids']]
        unary_ids = [f['id'] for f in g0.factors if len(f['vars']) == 1]
        for f in s['factors']:
            if len(f['vars']) == 1:
                f['observed_dim'] = int(obs[unary_ids.index(f['id'])])
        g = O.Graph(s)
        msgs = O.init_messages(g)
        roots = [s['var_ids'][i % len(s['var_ids'])] for i in range(3)]
        O.treelike_inference(g, inputs, msgs, 3, roots, O.has_loops(g, roots[0]))
        O.return_gradient(g, inputs, msgs, 0.2 / 8192.0, 0.1)
        O.log_posterior(g, msgs)
        n += 1
    return n


def cpu_train_epoch(directory, per_worker=512):
    """instances/s of the per-instance cost model on the host cores, on a bounded sample of the synthetic TI_DIR."""
    import multiprocessing as mp
    import numpy as np
    from macaronicusermodeling_amd import tidir
    paths = tidir.synthesize(directory, **TRAIN_EPOCH_TIDIR)
    en, de = tidir.read_vocab(paths['end']), tidir.read_vocab(paths['ded'])
    phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
    X = len(en)
    th_ee, th_ed = np.zeros((1, 3)), np.zeros((1, 6))              # the first epoch's thetas (train_mp.py:519-523)
    inputs = dict(phi_en_en=phi_ee, phi_en_en_w1=phi_w1, phi_en_de=phi_ed, theta_en_en=th_ee, theta_en_de=th_ed,
                  pot_en_en=np.exp(phi_ee.dot(th_ee.T).reshape(X, X)), pot_en_en_w1=np.exp(phi_w1.dot(th_ee.T).reshape(X, X)),
                  pot_en_de=np.exp(phi_ed.dot(th_ed.T).reshape(X, -1)))
    buckets = tidir.bucket_instances(tidir.read_instances(paths['ti']), en, de)
    specs = {str(k): b['spec'] for k, b in buckets.items()}
    flat = [(str(k), b['var_labels'][i], b['unary_obs'][i]) for k, b in buckets.items() for i in range(len(b['rows']))]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    jobs = [(specs, flat[w * per_worker:(w + 1) * per_worker], inputs) for w in range(cores)]
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_cpu_epoch_worker, [(specs, flat[:2], inputs)] * cores)      # imports and caches
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_epoch_worker, jobs))
        wall = time.perf_counter() - t0
    return paths, {'instances_per_s': done / wall, 'cores': cores, 'kind': 'port',
                   'sample': '%d instances of the same TI_DIR through the oracle\'s per-instance step (graph, initialize, 3 sweeps, '
                             'return_gradient, posterior: batch_sgd, train_mp.py:362-402), one process per core, %.1f s wall' % (done, wall)}


def gpu_train_epoch(paths, dev):
    """instances/s of TiDirTrainer.epoch on the synthetic TI_DIR: the whole file per update, and minibatches of 256 and 16 (the
    reference updates once per instance, train_mp.py:631-649) in the trainer's default ('masked') minibatch form, every step a
    HIP-graph replay."""
    import torch
    from macaronicusermodeling_amd.train import TiDirTrainer
    t0 = time.perf_counter()
    tt = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'],
                      device=dev, sweeps=3, minibatch=256, shuffle_seed=5)
    build_s = time.perf_counter() - t0
    # learning rate: the synthetic features are random, and a summed step of 8192 instances at the reference's 0.1 sends theta
    # (and with it every potential) out of range within two updates -- the run would time the degenerate-graph path.  1e-6 keeps
    # the optimisation in the regime a real run is in; the launches are the same.
    lr = 1e-6
    out = {'instances': tt.n_total, 'sentence_shapes': len(tt.trainers), 'trainer_build_s': build_s,
           'contents': 'TiDirTrainer.epoch: potentials, 3 sweeps, gradient, posterior, reduction, theta update (train_mp.py:626-656); '
                       'learning rate %g, reg 0.2 / N; every step one HIP-graph replay over the resident shard' % lr}
    reg = 0.2 / tt.n_total
    tt.capture_masked()
    for mb, epochs in ((256, 3), (16, 1)):
        tt.minibatch = mb
        tt.epoch(lr, reg)                       # (warm: allocator, first-use paths)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for e in range(epochs):
            tt.epoch(lr, reg)
        torch.cuda.synchronize(dev)
        el = (time.perf_counter() - t0) / epochs
        out['minibatch_%d' % mb] = {'instances_per_s': tt.n_total / el, 'ms_per_epoch': el * 1e3,
                                    'ms_per_update': el * 1e3 / ((tt.n_total + mb - 1) // mb), 'updates_per_epoch': (tt.n_total + mb - 1) // mb}
    tt.minibatch = None
    tt.capture()
    tt.epoch(lr, reg)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for e in range(5):
        tt.epoch(lr, reg)
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t0) / 5
    out['whole_file'] = {'instances_per_s': tt.n_total / el, 'ms_per_epoch': el * 1e3, 'updates_per_epoch': 1}
    out['graphs_redone_by_the_exact_kernel_in_the_last_step'] = int(sum(tr.batch.program(tr.roots[:tr.n_sweeps_run]).exact_count(tr.batch.B)
                                                                       for tr in tt.trainers.values()))
    return out


MIXED_EPOCH_TIDIR = dict(n_instances=8192, X=64, Vde=64, sent_len=(6, 9), n_predicted=(2, 4), seed=21)


def gpu_mixed_epoch(directory, dev):
    """The whole-file epoch on a TI_DIR whose sentences have 2-4 predicted words (train_mp.py:257-299 builds a K2 / K3 / K4 per
    instance): hundreds of sentence shapes of a dozen instances each, one sweep launch per form of the shared-table kernel (the
    product-fused one beside the three-source one on a side stream).  Secondary figure, never `value`."""
    import os as _os
    import torch
    from macaronicusermodeling_amd import tidir
    from macaronicusermodeling_amd.train import TiDirTrainer
    sub = _os.path.join(directory, 'mixed')
    _os.makedirs(sub, exist_ok=True)
    paths = tidir.synthesize(sub, **MIXED_EPOCH_TIDIR)
    t0 = time.perf_counter()
    tt = TiDirTrainer(paths['ti'], paths['end'], paths['ded'], paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'],
                      device=dev, sweeps=3)
    build_s = time.perf_counter() - t0
    lr, reg = 1e-6, 0.2 / tt.n_total
    by_p = {}
    for tr in tt.trainers.values():
        by_p[tr.topo.P] = by_p.get(tr.topo.P, 0) + tr.batch.B
    tt.capture()
    tt.epoch(lr, reg)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(10):
        tt.epoch(lr, reg)
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t0) / 10
    return {'instances': tt.n_total, 'sentence_shapes': len(tt.trainers), 'instances_by_pairwise_factors': {str(k): v for k, v in sorted(by_p.items())},
            'trainer_build_s': build_s, 'ms_per_epoch': el * 1e3, 'instances_per_s': tt.n_total / el, 'updates_per_epoch': 1,
            'contents': 'TiDirTrainer.epoch over the whole file, one HIP-graph replay per epoch; sentences with 2-4 predicted words'}


# ---------------------------------------------------------------------------------------------------
# CPU baseline (oracle = port of the reference's cost model).  Runs BEFORE any GPU initialisation.
# ---------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    spec, roots, seeds = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    import cases as C
    from oracle import lbp_oracle as O
    g = O.Graph(spec)
    n = 0
    for seed in seeds:
        inputs = C.make_inputs(spec, seed, 'uniform')
        t0 = time.perf_counter()
        msgs = O.init_messages(g)
        for r in roots:
            O.sweep(g, inputs, msgs, r)
        for v in g.var_order:
            O.marginal(g, msgs, v)
        n += 1
    return n


def cpu_baseline(workload, batch, budget_s=12.0):
    import multiprocessing as mp
    import cases as C
    from oracle import lbp_oracle as O
    spec, roots, sweeps, seed = workload_spec(workload)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))   # a 1-GPU box's CPU share
    # calibrate on one graph: the best of three passes (the first one also pays for lazy imports and cold caches)
    g = O.Graph(spec)
    inputs = C.make_inputs(spec, seed, 'uniform')
    t1 = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        msgs = O.init_messages(g)
        for r in roots:
            O.sweep(g, inputs, msgs, r)
        t1 = min(t1, max(time.perf_counter() - t0, 1e-4))
    per_worker = max(2, min(int(budget_s / t1 / 2), 4096))     # inputs cost about as much as sweeps
    ctx = mp.get_context('fork')
    with ctx.Pool(cores) as pool:
        for attempt in range(2):
            jobs = [(spec, roots, list(range(seed + w * per_worker, seed + (w + 1) * per_worker))) for w in range(cores)]
            t0 = time.perf_counter()
            done = sum(pool.map(_cpu_worker, jobs))
            wall = time.perf_counter() - t0
            if wall >= 0.4 * budget_s or per_worker >= 8192:
                break
            per_worker = min(8192, int(per_worker * 0.8 * budget_s / max(wall, 1e-3)))    # the sample was too short to mean much: once more, sized from it
    graph_sweeps_per_s = done * sweeps / wall
    return {'value': graph_sweeps_per_s / batch, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
            'sample': '%d graphs x %d sweeps of the same workload (%s), oracle/lbp_oracle.py, one process per '
                      'core, BLAS threads 1, %.1f s wall incl. synthetic input generation; value = graph-sweeps/s / %d'
                      % (done, sweeps, workload, wall, batch),
            'graph_sweeps_per_s': graph_sweeps_per_s}


# ---------------------------------------------------------------------------------------------------
def parity_sample(spec, topo, roots, fb, marg, n_graphs=4):
    """max relative marginal error of the GPU run against the CPU restatement in the same run, on the
    first few graphs of the batch (their tables are copied back; the oracle is the checker here, as in the
    cpu_baseline leg)."""
    import numpy as np
    from oracle import lbp_oracle as O
    X = spec['X']
    ex = dict(name=spec.get('name', 'bench'), style='explicit', X=X, var_ids=list(spec['var_ids']), labels=list(spec['labels']),
              factors=[dict(id=f['id'], vars=list(f['vars']), dims=list(f['dims']), table=f['id']) for f in spec['factors']])
    n_tab = 1 + max(f['id'] for f in spec['factors'])
    ptab = fb.pair_tab[:n_graphs].cpu().numpy() if topo.P else None
    utab = fb.unary_tab[:n_graphs].cpu().numpy() if topo.U else None
    worst = 0.0
    for b in range(min(n_graphs, fb.B)):
        tables = [None] * n_tab
        for p, j in enumerate(topo.pair_factors):
            tables[topo.factor_ids[j]] = fb.pair_tables[int(ptab[b, p])].double().cpu().numpy()
        for u, j in enumerate(topo.unary_factors):
            tables[topo.factor_ids[j]] = fb.unary_tables[int(utab[b, u])].cpu().numpy().reshape(X, 1)
        g = O.Graph(ex)
        msgs = O.init_messages(g)
        for r in roots:
            O.sweep(g, dict(tables=tables), msgs, r)
        got = marg[b].cpu().numpy()
        for k, v in enumerate(topo.var_ids):
            want = O.marginal(g, msgs, v).reshape(-1)
            worst = max(worst, float(np.max(np.abs(got[k] - want) / np.maximum(np.abs(want), 1e-300))))
    return {'max_rel_marginal_error': worst, 'graphs_compared': min(n_graphs, fb.B),
            'checker': 'oracle/lbp_oracle.py on the same tables (north star: <= 1e-5)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='user_k3', choices=['user_k3', 'user_k3_shared', 'user_k3_trainlayout', 'user_k4', 'user_k4_shared', 'user_k4_trainlayout', 'chain8', 'ring8', 'ring8_x512', 'ring8_x512_f32', 'ring8_x512_shared', 'ring8_x512_shared_f32', 'ring8_x1000', 'ring8_x1000_shared', 'ring8_x2048_shared'],
                    help='user_k3_shared = the same graphs with the reference\'s table layout: all graphs share the two '
                         'en_en pots (MFMA kernel, reported against the f64 matrix peak)')
    ap.add_argument('--no-writeback', action='store_true', help='shared workload: skip the message write-back (read-out only)')
    ap.add_argument('--batch', type=int, default=8192, help='graphs per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-skip-unchanged', action='store_true', help='leave out the secondary measurements -- MLBP_SWEEP_SKIP_UNCHANGED and the train step (profiling runs: only the full schedule of the sweep call is launched)')
    ap.add_argument('--sweeps', type=int, default=None, help='override sweeps per step (roots cycle)')
    ap.add_argument('--variant', type=int, default=None, help='mlbp_set_sweep_variant (A/B measurement)')
    ap.add_argument('--no-train-epoch', action='store_true', help='leave out the train_epoch figures (the outer loop on a synthetic TI_DIR)')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per sweep launch from a rocprofv3 --pmc pass (profiles/), if known')
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        return self_launch(a, sys.argv[1:])
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, a.gpus))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.workload, 8192)            # before the GPU is touched (fork safety)
    epoch_paths = epoch_cpu = epoch_dir = None
    if rank == 0 and world == 1 and not a.no_train_epoch and not a.no_skip_unchanged and a.workload == 'user_k3':
        import tempfile
        epoch_dir = tempfile.mkdtemp(prefix='mlbp_bench_tidir_')
        if a.no_cpu_baseline:
            from macaronicusermodeling_amd import tidir as _tidir
            epoch_paths = _tidir.synthesize(epoch_dir, **TRAIN_EPOCH_TIDIR)
        else:
            epoch_paths, epoch_cpu = cpu_train_epoch(epoch_dir)          # (also before the GPU is touched)

    import numpy as np
    import torch
    import torch.distributed as dist
    from macaronicusermodeling_amd.batch import FactorGraphBatch
    from macaronicusermodeling_amd.topology import GraphTopology

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU path')
    # MLBP_BENCH_BACKEND=gloo is a REHEARSAL switch for a box with fewer GPUs than ranks (ranks then
    # share devices and reduce through the host); the driver's real runs use RCCL ("nccl").
    backend = os.environ.get('MLBP_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    spec, roots, sweeps, seed = workload_spec(a.workload)
    if a.sweeps:
        roots = [roots[i % len(roots)] for i in range(a.sweeps)]
        sweeps = a.sweeps
    X, B = spec['X'], a.batch
    topo = GraphTopology.from_spec(spec)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 7919 * rank)
    fb = FactorGraphBatch(topo, X, B, device=dev)
    shared = '_shared' in a.workload or a.workload.endswith('_trainlayout')
    unary = torch.rand(B * topo.U, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
    unary_tab = None
    if a.workload.endswith('_trainlayout'):
        # the reference's real layout (train_mp.py:178-255, LBP.py:695-706): a unary factor's table is one column of a pot
        # shared by every instance under one theta -- 3 pots x 64 observed columns = 192 rows, stored transposed; which
        # row a factor reads is its observed word.  (UserGraphTrainer builds exactly this.)
        unary = torch.rand(3 * 64, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
        by_id = {f['id']: f for f in spec['factors']}
        kind = np.array([2 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else (0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1)
                         for j in topo.unary_factors])
        rs = np.random.RandomState(seed + rank)
        unary_tab = kind[None, :] * 64 + rs.randint(0, 64, size=(B, topo.U))
    if shared and spec['style'] == 'explicit':      # one table per factor, the same for every graph (X = 512: batched DGEMMs)
        pair = torch.rand(topo.P, X, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
        fb.set_pair_tables(pair, np.tile(np.arange(topo.P), (B, 1)), dtype=torch.float32 if a.workload.endswith('_f32') else torch.float64)
    elif shared:    # LBP.py:456-467: pot_en_en behind the gap > 1 factors, pot_en_en_w1 behind the gap == 1 ones
        pair = torch.rand(2, X, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
        by_id = {f['id']: f for f in spec['factors']}
        which = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
        fb.set_pair_tables(pair, np.tile(np.array(which), (B, 1)))
    elif a.workload.endswith('_f32'):   # BASELINE config 5: float32 tables (generated in float32: 2 x fewer bytes to hold, too)
        pair = torch.rand(B * topo.P, X, X, dtype=torch.float32, device=dev, generator=gen) + 0.01
        fb.set_pair_tables(pair, dtype=torch.float32)
    else:
        pair = torch.rand(B * topo.P, X, X, dtype=torch.float64, device=dev, generator=gen) + 0.01
        fb.set_pair_tables(pair)
    fb.set_unary_tables(unary, unary_tab)
    labels = np.tile(np.array([dict(zip(spec['var_ids'], spec['labels']))[v] for v in topo.var_ids]), (B, 1))
    labels_d = torch.from_numpy(labels.astype(np.int32)).to(dev)
    # two statistics buffers: the all-reduce of step i is asynchronous and overlaps step i+1's sweeps
    # (the reference's accumulate callback is asynchronous too, train_mp.py:405-424, 636-647)
    stats = [torch.zeros(16, dtype=torch.float64, device=dev) for _ in range(2)]
    pending = [None, None]
    step_no = [0]
    marg = torch.empty(B, topo.n_vars, X, dtype=torch.float64, device=dev)
    lp = torch.empty(B, dtype=torch.float64, device=dev)
    import ctypes as C
    from macaronicusermodeling_amd import _ffi
    from macaronicusermodeling_amd.batch import _stream_ptr

    if a.variant is not None:
        _ffi.check(_ffi.lib.mlbp_set_sweep_variant(a.variant))
    fb.initialize()
    fb.is_loopy = True          # chain workloads: run real sweeps, tree short-circuit overridden (LBP.py:219)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    submit = [0.0] * (a.steps + 1)

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        k = step_no[0] & 1
        step_no[0] += 1
        if pending[k] is not None:
            pending[k].wait()               # the all-reduce issued two steps ago (long finished)
        # initialize + marginal read-out fused into the launch; get_posterior_probs of every graph and their batch total
        # (train_mp.py:400, 405-411) ride on the call too -- the fix-up pass behind the fast kernel takes them -- and land
        # straight in the statistics buffer this step reduces
        fb.sweep(roots, init=True, marginals=marg, keep_messages=not a.no_writeback, posterior=(labels_d, lp, stats[k][:1]))
        if i is not None:
            ev[i][1].record()
        if world > 1:                       # the outer-loop reduction of train_mp.py:405-424: one per step
            pending[k] = dist.all_reduce(stats[k], async_op=True)

    # Device spin-up: after process start (and the CPU-baseline phase) the GPU has been idle for seconds and needs
    # ~100 launches to reach its steady clocks (measured: 0.32 ms per launch over the first 20, 0.29 ms after 300).
    # The same step runs untimed MLBP_BENCH_SPINUP_STEPS times (default 300, about 0.1 s; the same count on every
    # rank, the steps include the all-reduce) before the W warmup steps; set it to 0 to time from cold.
    step()                                  # lazy allocations / first-use module loads
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for _ in range(a.steps):                # the same K steps from a cold (idle-clock) device: reported as cold_ms_per_step
        step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - tc) / a.steps * 1e3
    # spin-up: MLBP_BENCH_SPINUP_STEPS steps (default 300) or MLBP_BENCH_SPINUP_SECONDS (default 1.0 s) worth of them,
    # whichever is more -- a device that has just come out of idle keeps changing power state for a while, and such a
    # change can stall it for tens of ms (seen once: one 40 ms launch inside a 20-step window, r02k).  The count is
    # derived from the cold step time agreed over the ranks, so every rank runs the same number; nothing synchronises
    # between here and the first window's own bracket, so the device does not fall idle again.
    import gc
    gc.collect()
    gc.disable()                            # no collector pause between two launches of a timed window
    spin_steps = int(os.environ.get('MLBP_BENCH_SPINUP_STEPS', '300'))
    if spin_steps > 0:
        c = torch.tensor([cold_ms], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(c, op=dist.ReduceOp.MAX)
        spin_steps = max(spin_steps, int(float(os.environ.get('MLBP_BENCH_SPINUP_SECONDS', '1.0')) * 1e3 / max(float(c.item()), 1e-3)))
    for _ in range(spin_steps):
        step()

    def timed_window():
        """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; returns
        (seconds, max over ranks), and the HIP events of its K sweep launches."""
        nonlocal ev, submit
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        submit = [0.0] * (a.steps + 1)
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            submit[i] = time.perf_counter()      # host clock at the launch's submission: tells a host stall from a device one
            step(i)
        submit[a.steps] = time.perf_counter()
        for w in pending:
            if w is not None:
                w.wait()                    # every step's reduction completes inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev, list(submit)

    # Three windows of exactly K steps each; the MEDIAN window is the one reported (value, ms_per_step, and the launch
    # times behind roofline): a single stalled launch (host pre-emption, a device power event) then cannot decide the
    # line either way.  All three are listed in `windows_ms_per_step`.
    windows = [timed_window() for _ in range(3)]
    gc.enable()
    order = sorted(range(3), key=lambda i: windows[i][0])
    elapsed, ev, submit = windows[order[1]]
    assert fb.program(roots).status() == 0

    # Secondary figure, never `value`: the same launch under MLBP_SWEEP_SKIP_UNCHANGED (include/mlbp.h), which drops the
    # updates of the root sequence that recompute a message from unchanged inputs.  `value` above executes every update
    # of the reference's schedule.  Rank 0 only, after the timed region, sweep launches alone (HIP events).
    skip = None
    if rank == 0:
        n_drop = fb.program(roots).skippable_updates()
        n_all = sum(len(topo.compile_sweep(r)[0]) for r in roots)
        if a.no_skip_unchanged:
            skip = None
        elif n_drop > 0:
            full = marg.clone()
            marg2 = torch.empty_like(marg)
            for _ in range(3):
                fb.sweep(roots, init=True, marginals=marg2, keep_messages=not a.no_writeback, skip_unchanged=True)
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            for _ in range(a.steps):
                fb.sweep(roots, init=True, marginals=marg2, keep_messages=not a.no_writeback, skip_unchanged=True)
            e_ev.record()
            torch.cuda.synchronize()
            skip_ms = s_ev.elapsed_time(e_ev) / a.steps
            skip = {'avg_launch_ms': skip_ms, 'updates_in_schedule': n_all, 'updates_dropped': n_drop,
                    'marginals_bit_identical_to_full_schedule': bool(torch.equal(full, marg2)),
                    'max_abs_marginal_difference': float((full - marg2).abs().max().item()),
                    'note': 'the launch without the updates whose inputs are unchanged since their destination was last '
                            'computed; not the headline: `value` executes all %d updates' % n_all}
        else:
            skip = {'updates_in_schedule': n_all, 'updates_dropped': 0}

    # SURVEY.md 8(d) config 3 reads "3 sweeps, then gradient" (LBP.py:301-327): the optimisation step of train_mp.py:381-400
    # on the same batch -- initialize + sweeps + per-graph gradient + log-posterior + the batch sums of batch_sgd_accumulate --
    # timed after the windows (HIP events, rank 0), never `value`.
    train = None
    if rank == 0 and spec['style'] == 'trainmp' and X == 64 and not a.no_skip_unchanged:      # (profiling runs launch the sweep call only)
        by_id = {f['id']: f for f in spec['factors']}
        pair_phi = [0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1 for j in topo.pair_factors]
        ukind = [2 if by_id[topo.factor_ids[j]]['factor_type'] == 'en_de' else (0 if by_id[topo.factor_ids[j]]['gap'] > 1 else 1)
                 for j in topo.unary_factors]
        rs = np.random.RandomState(seed + 5)
        Vde = spec['Vde']
        f_ee, f_w1 = rs.rand(X, X, 3), rs.rand(X, X, 3)
        if a.workload.endswith('_trainlayout'):      # the trainer's tensors as train_mp.py:600-606 stacks them: [pmi, 0, 1] and [pmi, pmi_w1, 1]
            f_ee[:, :, 1] = 0.0; f_ee[:, :, 2] = 1.0; f_w1[:, :, 0] = f_ee[:, :, 0]; f_w1[:, :, 2] = 1.0
        fb.set_features(f_ee, f_w1, rs.rand(X, Vde, 6), pair_phi, ukind)
        obs = np.stack([rs.randint(0, Vde if k == 2 else X, size=B) for k in ukind], axis=1)
        if a.workload.endswith('_trainlayout'):
            obs = unary_tab - np.array(ukind)[None, :] * 64           # the observed word IS the row a factor reads
        fb.set_observations(labels, obs)
        g_ee = torch.empty(B, 3, dtype=torch.float64, device=dev)
        g_ed = torch.empty(B, 6, dtype=torch.float64, device=dev)
        tstat = torch.zeros(16, dtype=torch.float64, device=dev)

        def train_step():
            fb.sweep(roots, init=True, marginals=marg, gradient=(g_ee, g_ed), keep_messages=False)
            _ffi.check(_ffi.lib.mlbp_log_posterior_f64(marg.data_ptr(), labels_d.data_ptr(), B, topo.n_vars, X, lp.data_ptr(), _stream_ptr(dev)))
            _ffi.check(_ffi.lib.mlbp_sum_rows_cat_f64(g_ee.data_ptr(), 3, g_ed.data_ptr(), 6, lp.data_ptr(), 1, B, 1, tstat.data_ptr(), _stream_ptr(dev)))
        for _ in range(5):
            train_step()
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        for _ in range(a.steps):
            train_step()
        e_ev.record()
        torch.cuda.synchronize()
        t_kernel = _ffi.lib.mlbp_last_sweep_kernel()
        fused = (g_ee.clone(), g_ed.clone())
        fb.sweep(roots, init=True, marginals=marg)                       # second opinion: the standalone gradient kernel on the
        sep = fb.gradient()                                              # messages this sweep leaves in memory
        train = {'ms': s_ev.elapsed_time(e_ev) / a.steps, 'kernel': KERNEL_NAMES.get(t_kernel, str(t_kernel)),
                 'contents': 'initialize + %d sweeps + get_unregularized_gradeint of every graph (LBP.py:301-320; fused into the sweep '
                             'launch when the kernel keeps the tables on chip) + get_posterior_probs + batch sums; messages not written back' % sweeps,
                 'max_abs_difference_to_standalone_gradient_kernel': float(max((fused[0] - sep[0]).abs().max().item(), (fused[1] - sep[1]).abs().max().item())),
                 'gradient_status': int(_ffi.lib.mlbp_gradient_status()), 'flagged_graphs': int(fb.program(roots).exact_count(B))}

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        iters_per_s = world * (B / 8192.0) * sweeps * a.steps / elapsed
        # the dominant kernel's launch time for the roofline: the sweep call ALONE (fast kernel + its fix-up pass, as in every round's
        # record), HIP events on the launch stream, K launches right behind the timed windows -- the step's own call also carries the
        # log-posteriors, which are not that kernel's bytes
        rev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for _ in range(3):
            fb.sweep(roots, init=True, marginals=marg, keep_messages=not a.no_writeback)
        for s_ev, e_ev in rev:
            s_ev.record()
            fb.sweep(roots, init=True, marginals=marg, keep_messages=not a.no_writeback)
            e_ev.record()
        torch.cuda.synchronize()
        sweep_ms = sorted(s_ev.elapsed_time(e_ev) for s_ev, e_ev in rev)
        avg_ms = sum(sweep_ms) / len(sweep_ms)
        table_elem = 4 if a.workload.endswith('_f32') else 8
        alg_bytes = algorithmic_bytes_per_graph(topo, roots, X, table_elem=table_elem) * B
        n_pt = int(torch.unique(fb.pair_tab).numel()) if topo.P else 0
        n_ur = int(torch.unique(fb.unary_tab).numel()) if topo.U else 0
        comp = compulsory_bytes(topo, X, B, n_pt, n_ur, table_elem, keep_messages=not a.no_writeback, init=True, marginals=True)
        last = _ffi.lib.mlbp_last_sweep_kernel()
        # which byte count prices the launch: the resident-table kernels (X <= 64) read every table once per launch;
        # the streaming kernels (wide / generic: a table of X^2 x 8 B per factor does not fit on chip) must re-read
        # it for every update, so for them SURVEY.md 8(d)'s per-update figure IS the compulsory traffic
        streamed = last in (4, 5)
        byte_model = ('streamed: every update re-reads its table (%d B per table and graph do not fit on chip)' % (X * X * table_elem)
                      if streamed else 'resident: every table read once per launch')
        priced = alg_bytes if streamed else comp['total']
        achieved = priced / (avg_ms * 1e-3) / 1e9
        # measured HBM traffic: only a rocprofv3 --pmc measurement of THIS workload made on THESE kernel sources counts
        sha = kernel_sources_sha()
        traffic, traffic_src = a.traffic_bytes, 'command line' if a.traffic_bytes else None
        if traffic is None:
            try:
                key = a.workload + ('_nowriteback' if a.no_writeback else '') + '_b%d' % B
                rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get(key)
                if not rec:
                    traffic_src = 'no rocprofv3 --pmc measurement of this workload in profiles/pmc_traffic.json'
                elif rec['batch'] != B or rec['sweeps'] != sweeps or a.variant not in (None, 1):
                    traffic_src = 'profiles/pmc_traffic.json holds this workload at another size or variant'
                elif rec.get('kernel_sources_sha') != sha:
                    traffic_src = 'stale: %s was measured on kernel sources %s, this run is %s' % (rec['source'], rec.get('kernel_sources_sha'), sha)
                else:
                    traffic, traffic_src = rec['hbm_bytes_per_launch'], rec['source']
            except (OSError, ValueError, KeyError) as e:
                traffic_src = 'profiles/pmc_traffic.json unreadable: %s' % e
        used_mfma = shared and last in (3, 6)
        if used_mfma:      # shared-table mode is priced in flops, 2 X^2 per pairwise update and graph
            n_pair = sum(int(np.isin(topo.compile_sweep(r)[0][:, 0], (_ffi.OP_PAIR_TM, _ffi.OP_PAIR_MT)).sum()) for r in roots)
            alg_flops = 2.0 * X * X * n_pair * B
            tfl = alg_flops / (avg_ms * 1e-3) / 1e12
            peak = F32_MFMA_PEAK_TFLOPS if a.workload.endswith('_f32') else F64_MFMA_PEAK_TFLOPS
            roof = {'bound': 'mfma', 'achieved': tfl, 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': tfl / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                    'kernel': KERNEL_NAMES.get(last, str(last)), 'algorithmic_flops_per_launch': alg_flops,
                    'hbm_GBps_on_compulsory_bytes': achieved, 'compulsory_bytes': comp}
        else:
            roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                    'kernel': KERNEL_NAMES.get(last, str(last)), 'byte_model': byte_model, 'compulsory_bytes': comp,
                    'algorithmic_bytes_per_launch': alg_bytes, 'algorithmic_GBps': alg_bytes / (avg_ms * 1e-3) / 1e9,
                    'table_reuse_factor': alg_bytes / comp['total']}
            if traffic:
                roof['frac_on_measured_traffic'] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if roof['frac'] > 1.0:      # only possible when re-reads were served on chip (a batch that fits the 256 MiB Infinity Cache)
                roof['note'] = 'fraction above 1: the re-read tables of this small batch were served by the Infinity Cache, not HBM'
        roof.update({'avg_launch_ms': avg_ms, 'min_launch_ms': sweep_ms[0], 'max_launch_ms': sweep_ms[-1],
                     'kernel_sources_sha': sha})
        # Where would a stall have come from?  Every window keeps, per step, the host time between two submissions and the
        # device time of the sweep launch (HIP events).  A step whose HOST gap is far above the median while its launch is
        # not is the host's (a pre-empted thread: the device idles, the events of the NEXT launch are unaffected); a launch
        # whose DEVICE time is far above the median is the device's (clock / power event) whatever the host did.
        def attribution(win):
            el, evs, sub = win
            gaps = [(sub[i + 1] - sub[i]) * 1e3 for i in range(a.steps)]
            devs = [s.elapsed_time(e) for s, e in evs]
            med_g, med_d = sorted(gaps)[len(gaps) // 2], sorted(devs)[len(devs) // 2]
            slow_host = [i for i in range(a.steps) if gaps[i] > 5 * med_g and gaps[i] > med_g + 1.0]
            slow_dev = [i for i in range(a.steps) if devs[i] > 5 * med_d and devs[i] > med_d + 1.0]
            return {'ms_per_step': el / a.steps * 1e3, 'median_host_submit_gap_ms': med_g, 'max_host_submit_gap_ms': max(gaps),
                    'median_launch_ms': med_d, 'max_launch_ms': max(devs),
                    'host_side_stalls': [{'step': i, 'gap_ms': gaps[i], 'launch_ms': devs[i]} for i in slow_host if i not in slow_dev],
                    'device_side_stalls': [{'step': i, 'gap_ms': gaps[i], 'launch_ms': devs[i]} for i in slow_dev]}
        stalls = [attribution(w) for w in windows]
        out = {
            'metric': 'LBP sweep iters/sec (whole node), batch=8192 graphs |X|=64',
            'value': iters_per_s, 'unit': 'iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'windows_ms_per_step': [w[0] / a.steps * 1e3 for w in windows], 'timing': 'median of three windows of exactly %d steps' % a.steps,
            'cold_ms_per_step': cold_ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 (f32 tables)' if a.workload.endswith('_f32') else 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: %d graphs/GPU, |X|=%d, P=%d pairwise + U=%d unary factors, unique %s '
                                   'table per (graph,factor)%s, step = initialize + %d sweeps + posterior read-out'
                                   % (a.workload, B, X, topo.P, topo.U, 'f32 pairwise / f64 unary' if a.workload.endswith('_f32') else 'f64', ' EXCEPT the pairwise tables: shared by all graphs' if shared else '', sweeps),
                       'graphs_per_gpu': B, 'X': X, 'sweeps_per_step': sweeps, 'roots': list(roots),
                       'graph_sweeps_per_s': world * B * sweeps * a.steps / elapsed,
                       'parallelism': 'graphs sharded over %d GPU(s), no data-path collective; one all-reduce of the '
                                      'step statistics per step (%s)' % (world, backend if world > 1 else 'n/a')},
            'roofline': roof,
            'cpu_baseline': cpu,
            'skip_unchanged': skip,
            'train_step': train,
            'stall_attribution': stalls,
        }
        if epoch_paths is not None:
            te = gpu_train_epoch(epoch_paths, dev)
            te['cpu_per_instance_cost_model'] = epoch_cpu
            try:
                te['mixed_shapes'] = gpu_mixed_epoch(epoch_dir, dev)
            except Exception as e:      # noqa: BLE001  (a secondary figure never takes the line down)
                te['mixed_shapes'] = {'error': repr(e)}
            out['train_epoch'] = te
            import shutil
            shutil.rmtree(epoch_dir, ignore_errors=True)
        if cpu is not None:
            out['parity'] = parity_sample(spec, topo, roots, fb, marg)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()          # rank 0 is still busy with the secondary figures and the parity sample: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
