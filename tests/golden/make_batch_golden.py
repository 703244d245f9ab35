#!/usr/bin/env python3
"""Pin the minibatched epoch and the prediction pass (train_mp.py:310-343, 405-424, 631-649)
on the reference's own code.

    python tests/golden/make_batch_golden.py [--reference /root/reference]

Same mechanism as make_tidir_golden.py: the named definitions of training_classes.py / train_mp.py are taken out of the
files' syntax trees (lib2to3 in memory) and executed against the reference's own LBP.py; nothing derived from the
reference's text is written to the repository.  Inputs: a synthetic TI_DIR from tidir.synthesize (12 instances, X = 64 --
`get_max_vocab(50)` needs 50 states -- V_de = 12), seeded non-zero theta, all three feature planes on.

Saved in tidir_batch_reference.json:
  * the TI_DIR (instances, vocabularies, feature matrices) and theta;
  * `minibatch`: the instances in a fixed shuffled order, cut into minibatches of 4; every instance of a minibatch goes
    through `batch_sgd` at the theta the minibatch starts from and the returned steps are added to theta as
    `batch_sgd_accumulate` adds them (train_mp.py:419-424) -- theta after each minibatch, the per-instance log-posteriors;
  * `predictions`: every instance through `batch_predictions` (train_mp.py:310-343, qp=False) in file order: the
    '*SENT_ID:' block, the .dist lines, get_posterior_probs and the precision counts -- the text a prediction run writes
    with --save_predictions (train_mp.py:740-760).
Needs /root/reference; never run on the GPU box."""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as G  # noqa: E402
from make_tidir_golden import definitions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    a = ap.parse_args()
    from macaronicusermodeling_amd import tidir
    L, au, cleanup = G.load_reference(a.reference)
    tmp = tempfile.mkdtemp(prefix='mlbp_batch_')
    try:
        roots = G.Roots(L)
        L.FactorNode.__lt__ = lambda self, other: self.id < other.id       # Python 2 ordered any two objects (make_text_golden.py)
        tc = {'sys': sys}
        exec(definitions(os.path.join(a.reference, 'training_classes.py'), ['TrainingInstance', 'Guess', 'SimpleNode']), tc)
        X, Vde, n_inst = 64, 12, 12
        paths = tidir.synthesize(tmp, n_instances=n_inst, X=X, Vde=Vde, sent_len=(4, 7), n_predicted=(1, 3), seed=12)
        lines = [l for l in open(paths['ti'], encoding='utf8').read().split('\n') if l.strip()]
        en, de = tidir.read_vocab(paths['end']), tidir.read_vocab(paths['ded'])
        phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
        rs = np.random.RandomState(6)
        ee_names, ed_names = ['pmi', 'pmi_w1', 'bias'], ['ed', 'ped', 'correct', 'full_history', 'hit_history', 'bias']     # train_mp.py:520-522
        theta_ee, theta_ed = rs.randn(1, 3) * 0.5, rs.randn(1, 6) * 0.5
        ns = {'np': np, 'sys': sys, 'json': json, 'DTYPE': np.float64, 'PRED2GIVEN': 'pred2given', 'PRED2PRED': 'pred2pred',
              'VariableNode': L.VariableNode, 'FactorNode': L.FactorNode, 'FactorGraph': L.FactorGraph, 'PotentialTable': L.PotentialTable,
              'VAR_TYPE_GIVEN': L.VAR_TYPE_GIVEN, 'VAR_TYPE_PREDICTED': L.VAR_TYPE_PREDICTED, 'TrainingInstance': tc['TrainingInstance'],
              'options': types.SimpleNamespace(user_adapt=False, experience_adapt=False, use_correct_feat=True, history=True,
                                               session_history=True, use_approx_beliefs=False, use_approx_inference=False,
                                               report_times=False, reg_param=0.2, reg_param_ua_scale='1.0'),
              'N': len(lines), 'de_domain': de, 'domain2theta': {}}
        exec(definitions(os.path.join(a.reference, 'train_mp.py'),
                         ['find_guess', 'get_var_node_pair', 'create_factor_graph', 'apply_regularization', 'batch_sgd', 'batch_predictions']), ns)
        en2id = {w: i for i, w in enumerate(en)}
        de2id = {w: i for i, w in enumerate(de)}

        def root_queue(line):
            """has_loops' start, then one root per sweep the reference will run: the predicted positions in order, cyclic
            (the batched trainer's rule); a tree runs one sweep (LBP.py:219)."""
            rec = json.loads(line)
            sent = sorted(rec['current_sent'], key=lambda n: n['position'])
            guessed = {tuple(g['id']) for g in rec['current_guesses']}
            vids = [i for i, n in enumerate(sent) if n['lang'] != 'en' and tuple(n['id']) in guessed]
            loopy = len(vids) >= 3
            return [vids[0]] + [vids[i % len(vids)] for i in range(3 if loopy else 1)], vids

        err = sys.stderr
        # ---- minibatched epoch ----
        order = [int(v) for v in np.random.RandomState(7).permutation(len(lines))]
        k, lr = 4, 0.1
        th_ee, th_ed = theta_ee.copy(), theta_ed.copy()
        mini = []
        sys.stderr = open(os.devnull, 'w')
        try:
            for m0 in range(0, len(order), k):
                idx = order[m0:m0 + k]
                acc_ee, acc_ed, logps = np.zeros_like(th_ee), np.zeros_like(th_ed), []
                for i in idx:
                    roots.queue, _ = root_queue(lines[i])
                    phi = L.PhiWrapper(phi_ee.copy(), phi_w1.copy(), phi_ed.copy())
                    sent_id, p, g_ee, g_ed, ag = ns['batch_sgd'](lines[i], ee_names, ed_names, th_ee.copy(), th_ed.copy(), phi, lr, en, de2id, en2id, {})
                    assert not roots.queue
                    acc_ee += g_ee; acc_ed += g_ed                       # batch_sgd_accumulate, train_mp.py:419-424
                    logps.append(float(np.sum(p)))
                th_ee, th_ed = th_ee + acc_ee, th_ed + acc_ed
                mini.append(dict(instances=idx, log_posteriors=logps, theta_en_en=th_ee.reshape(-1).tolist(), theta_en_de=th_ed.reshape(-1).tolist()))
        finally:
            sys.stderr = err
        # ---- prediction pass ----
        preds = []
        sys.stderr = open(os.devnull, 'w')
        try:
            for line in lines:
                roots.queue, _ = root_queue(line)
                phi = L.PhiWrapper(phi_ee.copy(), phi_w1.copy(), phi_ed.copy())
                p, fgs, dist, prec = ns['batch_predictions'](line, ee_names, ed_names, theta_ee.copy(), theta_ed.copy(), phi, 0.05, en, de2id, en2id, {})
                assert not roots.queue
                preds.append(dict(log_posterior=float(np.sum(p)), block=fgs, dist=dist, precision=[int(v) for v in prec]))
        finally:
            sys.stderr = err
        out = dict(X=X, Vde=Vde, vocab_en=en, vocab_de=de, instances=lines,
                   phi_pmi=np.loadtxt(paths['phi_pmi']).tolist(), phi_pmi_w1=np.loadtxt(paths['phi_pmi_w1']).tolist(),
                   phi_ed=np.loadtxt(paths['phi_ed']).tolist(), phi_ped=np.loadtxt(paths['phi_ped']).tolist(),
                   theta_en_en=theta_ee.tolist(), theta_en_de=theta_ed.tolist(), ee_names=ee_names, ed_names=ed_names,
                   options=dict(use_correct_feat=True, history=True, session_history=True, sweeps=3, reg_param=0.2),
                   minibatch=dict(order=order, size=k, learning_rate=lr, steps=mini),
                   predictions=preds)
        json.dump(out, open(os.path.join(HERE, 'tidir_batch_reference.json'), 'w'), ensure_ascii=False)
        man_path = os.path.join(HERE, 'MANIFEST.json')
        man = json.load(open(man_path)) if os.path.exists(man_path) else {}
        man['tidir_batch_reference'] = {'generator': 'tests/golden/make_batch_golden.py',
                                        'reference_files': {f: hashlib.sha256(open(os.path.join(a.reference, f), 'rb').read()).hexdigest()
                                                            for f in ('training_classes.py', 'train_mp.py', 'LBP.py')}}
        json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
        print('wrote tidir_batch_reference.json: %d minibatches, %d prediction blocks' % (len(mini), len(preds)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        cleanup()


if __name__ == '__main__':
    main()
