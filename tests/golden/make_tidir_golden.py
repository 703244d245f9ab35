#!/usr/bin/env python3
"""Pin the TI_DIR reader and the sentence -> factor-graph compiler on the reference's own code.

    python tests/golden/make_tidir_golden.py [--reference /root/reference]

`training_classes.py` imports `enchant` (absent) and `train_mp.py` imports `training_classes`, so neither module can be
imported; both are Python 2.  What the path needs from them uses neither enchant nor the edit-distance helper:
the dict-classes TrainingInstance / Guess / SimpleNode (training_classes.py:8-39, 94-183) and find_guess /
get_var_node_pair / create_factor_graph / apply_regularization / batch_sgd (train_mp.py:34-47, 105-306, 360-398).  This script reads the two files as text, passes them
through the stdlib `lib2to3` fixers IN MEMORY (as make_golden.py does for LBP.py), takes exactly those definitions out of the
syntax trees and executes them against the reference's own LBP.py (loaded by make_golden.load_reference), with the
module-level names the functions read (`options`, `N`, `de_domain`, PRED2PRED / PRED2GIVEN -- set from the command line in
train_mp.py's __main__) given as inputs.  Nothing derived from the reference's text is written to the repository.

Inputs: a synthetic TI_DIR from tidir.synthesize (12 instances, X = 16, V_de = 12), a few guesses re-spelt so that the
normalisation rules act (phrasal guess, trailing '*', apostrophe, capitals), seeded non-zero theta, all three feature
planes on.  Saved in tidir_reference.json: the TI_DIR itself (instances, vocabularies, the four feature matrices), theta,
and per instance what the reference built and computed -- normalised guesses and nodes, variables (type, label, truth),
factors in creation order (type, variables, observed index, gap, position, word label), the root sequence used, marginals
after initialize + three sweeps, get_posterior_probs, and the instance's step `return_gradient()` (learning rate 0.1,
regularisation options.reg_param / N as train_mp.py:160 sets it; what batch_sgd returns and batch_sgd_accumulate adds to theta).  Plus `user_adapt`: the same instances through batch_sgd (train_mp.py:360-398) with --user_adapt on and seeded per-user
thetas: log-posterior, global step and per-domain step of every instance; `experience_adapt`: the same with --experience_adapt
(domain = number of sentences seen).  Plus `normalisation`: raw -> Guess.guess / SimpleNode fields for
a list of spellings.  Needs /root/reference; never run on the GPU box."""
import argparse
import ast
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


def definitions(path, names):
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(open(path).read() + '\n', os.path.basename(path))))
    got = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in got) == sorted(names), [n.name for n in got]
    return compile(ast.Module(body=got, type_ignores=[]), path, 'exec')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    a = ap.parse_args()
    from macaronicusermodeling_amd import tidir
    L, au, cleanup = G.load_reference(a.reference)
    tmp = tempfile.mkdtemp(prefix='mlbp_tidir_')
    try:
        roots = G.Roots(L)
        L.FactorNode.__lt__ = lambda self, other: self.id < other.id       # Python 2 ordered any two objects (make_text_golden.py)
        tc = {'sys': sys}
        exec(definitions(os.path.join(a.reference, 'training_classes.py'), ['TrainingInstance', 'Guess', 'SimpleNode']), tc)
        X, Vde = 16, 12
        paths = tidir.synthesize(tmp, n_instances=12, X=X, Vde=Vde, sent_len=(4, 7), n_predicted=(1, 3), seed=11)
        lines = [l for l in open(paths['ti'], encoding='utf8').read().split('\n') if l.strip()]
        # re-spell a few guesses: same word after Guess.__init__'s rules
        recs = [json.loads(l) for l in lines]
        respell = [lambda g: 'a ' + g.upper() + '*', lambda g: g[:2] + "'" + g[2:], lambda g: ' ' + g.capitalize() + ' ', lambda g: g + '*']
        k = 0
        for idx, r in enumerate(recs):
            r['past_sentences_seen'] = list(range(idx % 3))          # --experience_adapt: domain = how many sentences were seen
            for fld in ('current_guesses', 'current_revealed_guesses', 'past_correct_guesses', 'past_guesses_for_current_sent'):
                for g in r[fld]:
                    if k % 3 == 0:
                        g['guess'] = respell[(k // 3) % len(respell)](g['guess'])
                    k += 1
        lines = [json.dumps(r) for r in recs]
        en, de = tidir.read_vocab(paths['end']), tidir.read_vocab(paths['ded'])
        phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
        rs = np.random.RandomState(5)
        ee_names, ed_names = ['pmi', 'pmi_w1', 'bias'], ['ed', 'ped', 'correct', 'full_history', 'hit_history', 'bias']     # train_mp.py:520-522
        theta_ee, theta_ed = rs.randn(1, 3) * 0.7, rs.randn(1, 6) * 0.7
        ns = {'np': np, 'sys': sys, 'DTYPE': np.float64, 'PRED2GIVEN': 'pred2given', 'PRED2PRED': 'pred2pred',
              'VariableNode': L.VariableNode, 'FactorNode': L.FactorNode, 'FactorGraph': L.FactorGraph, 'PotentialTable': L.PotentialTable,
              'VAR_TYPE_GIVEN': L.VAR_TYPE_GIVEN, 'VAR_TYPE_PREDICTED': L.VAR_TYPE_PREDICTED,
              'options': types.SimpleNamespace(user_adapt=False, experience_adapt=False, use_correct_feat=True, history=True,
                                               session_history=True, use_approx_beliefs=False, use_approx_inference=False,
                                               report_times=False, reg_param=0.1),
              'N': len(lines), 'de_domain': de}
        exec(definitions(os.path.join(a.reference, 'train_mp.py'), ['find_guess', 'get_var_node_pair', 'create_factor_graph']), ns)
        en2id = {w: i for i, w in enumerate(en)}
        de2id = {w: i for i, w in enumerate(de)}
        out_inst = []
        err = sys.stderr
        sys.stderr = open(os.devnull, 'w')          # create_factor_graph writes a progress dot per instance
        try:
            for line in lines:
                ti = tc['TrainingInstance'].from_dict(json.loads(line))
                phi = L.PhiWrapper(phi_ee.copy(), phi_w1.copy(), phi_ed.copy())
                fg = ns['create_factor_graph'](ti=ti, learning_rate=0.1, theta_en_en_names=ee_names, theta_en_de_names=ed_names,
                                               theta_en_en=theta_ee.copy(), theta_en_de=theta_ed.copy(), phi_wrapper=phi, en_domain=en,
                                               de2id=de2id, en2id=en2id, d2t={})
                vids = sorted(fg.variables.keys())
                seq = [vids[i % len(vids)] for i in range(3)]
                roots.queue = [vids[0]]
                fg.initialize()
                fg.isLoopy = True                   # run the three sweeps also on a tree (the batched trainer always does)
                roots.queue = list(seq)
                fg.treelike_inference(3)
                rec = dict(
                    guesses={fld: [[list(g.id), g.guess, bool(g.revealed), g.l2_word, g.reference] for g in getattr(ti, fld)]
                             for fld in ('current_guesses', 'current_revealed_guesses', 'past_correct_guesses', 'past_guesses_for_current_sent')},
                    nodes=[[n.sent_id, list(n.id), n.l2_word, n.l1_parent, n.position, n.lang] for n in ti.current_sent],
                    variables=[[v, fg.variables[v].var_type, fg.variables[v].supervised_label, fg.variables[v].truth_label] for v in vids],
                    factors=[[f.id, f.factor_type, [v.id for v in f.varset], f.potential_table.observed_dim, f.gap, f.position, f.word_label]
                             for f in sorted(fg.factors, key=lambda f: f.id)],
                    roots=seq,
                    marginals=[fg.variables[v].get_marginal().m.reshape(-1).tolist() for v in vids],
                    # batch_sgd's result (train_mp.py:398): the learning-rate-scaled regularised step of this instance
                    step=[np.asarray(g, dtype=np.float64).reshape(-1).tolist() for g in fg.return_gradient()],
                    log_posterior=float(np.sum(fg.get_posterior_probs())), log_posterior_terms=np.asarray(fg.get_posterior_probs(), dtype=np.float64).reshape(-1).tolist())
                out_inst.append(rec)
        finally:
            sys.stderr = err
        # ---- --user_adapt (train_mp.py:162-171, 226-247, 382-394): potentials from the user's theta INSTEAD of the global one,
        #      batch_sgd returns the global step and the per-domain step (regularisation scaled by reg_param_ua_scale) ----
        users = sorted({json.loads(l)['user_id'] for l in lines})
        d2t = {}
        for u in users:
            d2t['en_en', u] = rs.randn(1, 3) * 0.7
            d2t['en_de', u] = rs.randn(1, 6) * 0.7
        ns['options'].user_adapt = True
        ns['options'].reg_param_ua_scale = '0.5'
        ns['domain2theta'] = d2t
        ns['json'] = json
        ns['TrainingInstance'] = tc['TrainingInstance']
        exec(definitions(os.path.join(a.reference, 'train_mp.py'), ['apply_regularization', 'batch_sgd']), ns)
        theta_dom0 = {u: [d2t['en_en', u].reshape(-1).tolist(), d2t['en_de', u].reshape(-1).tolist()] for u in users}
        adapt_inst = []
        sys.stderr = open(os.devnull, 'w')
        try:
            for line, rec in zip(lines, out_inst):
                vids = [v[0] for v in rec['variables']]
                roots.queue = [vids[0]] + [vids[i % len(vids)] for i in range(3)]       # has_loops' start, then the sweeps' roots
                phi = L.PhiWrapper(phi_ee.copy(), phi_w1.copy(), phi_ed.copy())
                sent_id, p, g_ee, g_ed, ag = ns['batch_sgd'](line, ee_names, ed_names, theta_ee.copy(), theta_ed.copy(), phi, 0.1, en, de2id, en2id,
                                                              {k: v.copy() for k, v in d2t.items()})
                (u,) = {d for _, d in ag}
                adapt_inst.append(dict(sent_id=sent_id, user=u, log_posterior=float(np.sum(p)),
                                       step=[np.asarray(g_ee).reshape(-1).tolist(), np.asarray(g_ed).reshape(-1).tolist()],
                                       step_domain=[np.asarray(ag['en_en', u]).reshape(-1).tolist(), np.asarray(ag['en_de', u]).reshape(-1).tolist()]))
        finally:
            sys.stderr = err
            roots.queue = []
        # ---- --experience_adapt (train_mp.py:167-171): the same with domain = len(ti.past_sentences_seen) ----
        seen = sorted({len(json.loads(l)['past_sentences_seen']) for l in lines})
        d2t_e = {}
        for d in seen:
            d2t_e['en_en', d] = rs.randn(1, 3) * 0.7
            d2t_e['en_de', d] = rs.randn(1, 6) * 0.7
        ns['options'].user_adapt = False
        ns['options'].experience_adapt = True
        ns['options'].reg_param_ua_scale = '2.0'
        ns['domain2theta'] = d2t_e
        theta_exp0 = {str(d): [d2t_e['en_en', d].reshape(-1).tolist(), d2t_e['en_de', d].reshape(-1).tolist()] for d in seen}
        exp_inst = []
        sys.stderr = open(os.devnull, 'w')
        try:
            for line, rec in zip(lines, out_inst):
                vids = [v[0] for v in rec['variables']]
                roots.queue = [vids[0]] + [vids[i % len(vids)] for i in range(3)]
                phi = L.PhiWrapper(phi_ee.copy(), phi_w1.copy(), phi_ed.copy())
                sent_id, p, g_ee, g_ed, ag = ns['batch_sgd'](line, ee_names, ed_names, theta_ee.copy(), theta_ed.copy(), phi, 0.1, en, de2id, en2id,
                                                              {k: v.copy() for k, v in d2t_e.items()})
                (d,) = {d for _, d in ag}
                exp_inst.append(dict(sent_id=sent_id, domain=str(d), log_posterior=float(np.sum(p)),
                                     step=[np.asarray(g_ee).reshape(-1).tolist(), np.asarray(g_ed).reshape(-1).tolist()],
                                     step_domain=[np.asarray(ag['en_en', d]).reshape(-1).tolist(), np.asarray(ag['en_de', d]).reshape(-1).tolist()]))
        finally:
            sys.stderr = err
            roots.queue = []
        raw_guesses = ['', '   ', '__BLANK__', '__blank__', '__Unk__', '__copy__', 'House', "don't", 'the big house', 'big* house', 'star*', '*',
                       ' x ', 'Ab Cd*', "o'neil's*", 'aa bb', 'bb aa', 'ünï Code']
        norm_g = [[r, tc['Guess'](id=(0, 0), guess=r, revealed=False, l2_word='w').guess] for r in raw_guesses]
        raw_nodes = [dict(sent_id=1, id=[1, 2], l2_word="Don't", l1_parent="It's", position='3', lang='en'),
                     dict(sent_id=1, id=[1, 3], l2_word="Straße'N", l1_parent="The Street's", position=4, lang='de')]
        norm_n = []
        for d in raw_nodes:
            n = tc['SimpleNode'].from_dict(d)
            norm_n.append([d, [n.sent_id, list(n.id), n.l2_word, n.l1_parent, n.position, n.lang]])
        out = dict(X=X, Vde=Vde, vocab_en=en, vocab_de=de, instances=lines,
                   phi_pmi=np.loadtxt(paths['phi_pmi']).tolist(), phi_pmi_w1=np.loadtxt(paths['phi_pmi_w1']).tolist(),
                   phi_ed=np.loadtxt(paths['phi_ed']).tolist(), phi_ped=np.loadtxt(paths['phi_ped']).tolist(),
                   theta_en_en=theta_ee.tolist(), theta_en_de=theta_ed.tolist(), ee_names=ee_names, ed_names=ed_names,
                   options=dict(use_correct_feat=True, history=True, session_history=True, sweeps=3, learning_rate=0.1, reg_param=0.1),
                   reference=out_inst, normalisation=dict(guesses=norm_g, nodes=norm_n),
                   user_adapt=dict(users=users, theta_dom=theta_dom0, reg_param_ua_scale=0.5, instances=adapt_inst),
                   experience_adapt=dict(domains=[str(d) for d in seen], theta_dom=theta_exp0, reg_param_ua_scale=2.0, instances=exp_inst))
        json.dump(out, open(os.path.join(HERE, 'tidir_reference.json'), 'w'), ensure_ascii=False)
        man_path = os.path.join(HERE, 'MANIFEST.json')
        man = json.load(open(man_path)) if os.path.exists(man_path) else {}
        man['tidir_reference'] = {'generator': 'tests/golden/make_tidir_golden.py',
                                  'reference_files': {f: hashlib.sha256(open(os.path.join(a.reference, f), 'rb').read()).hexdigest()
                                                      for f in ('training_classes.py', 'train_mp.py', 'LBP.py')}}
        json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
        print('wrote tidir_reference.json:', len(out_inst), 'instances; shapes',
              sorted({(len(r['nodes']), len(r['variables'])) for r in out_inst}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        cleanup()


if __name__ == '__main__':
    main()
