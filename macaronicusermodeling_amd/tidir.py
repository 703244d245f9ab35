"""Data formats either side of the hot path -- host-side only.

* training-instance files: one JSON object per line with the `TrainingInstance` schema of
  training_classes.py:8-39 (`Guess` 94-147, `SimpleNode` 150-183), including the reference's
  normalisation rules for guesses and words;
* vocabulary files (one token per line, train_mp.py:591-594) and feature matrices in `np.loadtxt`
  text form stacked into the phi tensors exactly as train_mp.py:597-621 does;
* the instance -> factor-graph-shape compiler: `get_var_node_pair` + the factor loops of
  `create_factor_graph` (train_mp.py:105-133, 257-299) as integer work, emitting one *bucket* per
  sentence shape (length + predicted positions) with the per-instance label / observation arrays the
  batched engine consumes (`train.UserGraphTrainer`);
* the params file (`save_params` / `read_params`, train_mp.py:49-102) and the `*SENT_ID:` prediction
  blocks (train_mp.py:337) built on `FactorGraph.to_string`.

A synthetic TI_DIR generator with the same schema stands in for the reference's data, which is not in
its repository (README:3).  The per-instance feature planes `correct` / `full_history` /
`hit_history` (train_mp.py:178-217) are parsed into sparse (i, j, value) lists per instance and applied
by the batched trainer through private table rows (mlbp_patch_unary_tables_f64 / mlbp_patch_gradient_f64).
"""
import codecs
import json
import os

import numpy as np

EE_NAMES = ['pmi', 'pmi_w1', 'bias']                                              # train_mp.py:520
ED_NAMES = ['ed', 'ped', 'correct', 'full_history', 'hit_history', 'bias']       # train_mp.py:522


# ---- normalisation rules of the reference's dict-objects ------------------------------------------
def normalize_guess(guess):
    """Guess.__init__, training_classes.py:97-110."""
    g = guess.strip()
    if g == '':
        return '__blank__'
    if g.lower() in ('__blank__', '__unk__', '__copy__'):
        return guess
    g = sorted([(len(w), w) for w in guess.split()])[-1][1]          # phrasal guess -> its longest token
    if g[-1] == '*' and len(g) > 1:
        g = g[:-1]
    return g.lower().replace("'", "")


def parse_guess(d):
    return dict(id=tuple(d['id']), guess=normalize_guess(d['guess']), revealed=d['revealed'], l2_word=d['l2_word'],
                reference=d.get('reference'))


def parse_node(d):
    """SimpleNode.__init__, training_classes.py:151-165."""
    lang = d['lang']
    l2 = d['l2_word'].lower().replace("'", "") if lang == 'en' else d['l2_word']
    l1 = d['l1_parent'].lower().replace("'", "") if lang == 'de' else d['l1_parent']
    return dict(sent_id=d['sent_id'], id=tuple(d['id']), l2_word=l2, l1_parent=l1, position=int(d['position']), lang=lang)


def parse_instance(line):
    """TrainingInstance.from_dict, training_classes.py:29-39."""
    d = json.loads(line) if isinstance(line, str) else line
    return dict(user_id=d['user_id'],
                past_correct_guesses=[parse_guess(g) for g in d['past_correct_guesses']],
                past_sentences_seen=d['past_sentences_seen'],
                past_guesses_for_current_sent=[parse_guess(g) for g in d['past_guesses_for_current_sent']],
                current_sent=[parse_node(n) for n in d['current_sent']],
                current_revealed_guesses=[parse_guess(g) for g in d['current_revealed_guesses']],
                current_guesses=[parse_guess(g) for g in d['current_guesses']])


def read_instances(path):
    with codecs.open(path, 'r', 'utf8') as f:
        return [parse_instance(l) for l in f if l.strip()]


def read_vocab(path):
    with codecs.open(path, 'r', 'utf8') as f:
        return [l.strip() for l in f]


def load_features(phi_pmi, phi_pmi_w1, phi_ed, phi_ped):
    """-> (phi_en_en, phi_en_en_w1, phi_en_de) stacked as train_mp.py:597-621."""
    pmi = np.loadtxt(phi_pmi, dtype=np.float64, ndmin=2)
    w1 = np.loadtxt(phi_pmi_w1, dtype=np.float64, ndmin=2)
    bias = np.ones_like(pmi)
    phi_en_en_w1 = np.stack([pmi, w1, bias], axis=2)
    phi_en_en = np.stack([pmi, np.zeros_like(pmi), bias], axis=2)
    ed = np.loadtxt(phi_ed, dtype=np.float64, ndmin=2)
    ped = np.loadtxt(phi_ped, dtype=np.float64, ndmin=2)
    z = np.zeros_like(ed)
    phi_en_de = np.stack([ed, ped, z, z.copy(), z.copy(), np.ones_like(ed)], axis=2)
    return phi_en_en, phi_en_en_w1, phi_en_de


# ---- instance -> shape compiler ---------------------------------------------------------------------
def _find(node_id, guesses):
    for g in guesses:
        if g['id'] == node_id:
            return g
    return None


def instance_shape(ti, en2id, de2id):
    """One instance -> (shape key, per-instance arrays).  Variable id = index in position order;
    en nodes and revealed words are GIVEN, guessed words PREDICTED (train_mp.py:105-133)."""
    sent = sorted(ti['current_sent'], key=lambda n: n['position'])
    predicted, label, de_obs = [], [], []
    for idx, n in enumerate(sent):
        if n['lang'] == 'en':
            label.append(en2id[n['l2_word']]); de_obs.append(-1)
            continue
        g = _find(n['id'], ti['current_guesses'])
        if g is None:
            g = _find(n['id'], ti['current_revealed_guesses'])
            assert g is not None
        else:
            predicted.append(idx)
        label.append(en2id[g['guess']])
        de_obs.append(de2id[n['l2_word']])
    planes = dict(correct=[], full_history=[], hit_history=[])
    for cg in ti['current_guesses']:                                     # train_mp.py:180-186
        if cg['guess'] == cg['reference']:
            planes['correct'].append((en2id[cg['guess']], de2id[cg['l2_word']], 1.0))
    for pg in ti['past_correct_guesses']:                                # train_mp.py:194-199
        planes['full_history'].append((en2id[pg['guess']], de2id[pg['l2_word']], 1.0))
    for ig in ti['past_guesses_for_current_sent']:                       # train_mp.py:208-212
        if not ig['revealed']:
            planes['hit_history'].append((en2id[ig['guess']], de2id[ig['l2_word']], -1.0))
    # what the prediction text needs beside the numbers (LBP.py:109-143): per position the word a line starts with -- the l2
    # word under a predicted variable (FactorNode.word_label, train_mp.py:264), the label of a given one (train_mp.py:298) --
    # and a predicted variable's truth (SimpleNode.l1_parent, train_mp.py:108, 128)
    words = [n['l2_word'] if i in predicted else None for i, n in enumerate(sent)]
    truth = [n['l1_parent'] if i in predicted else None for i, n in enumerate(sent)]
    return (len(sent), tuple(predicted)), dict(label=label, de_obs=de_obs, planes=planes,
                                               sent_id=ti['current_sent'][0]['sent_id'], user_id=ti['user_id'],
                                               n_seen=len(ti['past_sentences_seen']), words=words, truth=truth)


def shape_spec(sent_len, predicted, X, Vde, name=None):
    """The factor list create_factor_graph builds for a sentence shape (train_mp.py:257-299): per
    predicted var one unary en_de factor (gap 0); per predicted pair one pairwise en_en (gap = |id
    diff|, dim 0 = the earlier word); per (predicted, given) pair one unary en_en factor.  Observed
    columns are per instance (`unary_obs`), so `observed_dim` is a placeholder here."""
    predicted = sorted(predicted)
    factors = []
    for i in predicted:
        factors.append(dict(id=len(factors), vars=[i], dims=[0], factor_type='en_de', gap=0, observed_dim=0,
                            obs_size=Vde, position=i, given=None))
    for a in range(sent_len):
        for b in range(a + 1, sent_len):
            pa, pb = a in predicted, b in predicted
            if pa and pb:
                factors.append(dict(id=len(factors), vars=[a, b], dims=[0, 1], factor_type='en_en', gap=abs(a - b),
                                    observed_dim=None, obs_size=None, position=None, given=None))
            elif pa or pb:
                g, p = (a, b) if pb else (b, a)
                factors.append(dict(id=len(factors), vars=[p], dims=[0], factor_type='en_en', gap=abs(g - p),
                                    observed_dim=0, obs_size=X, position=g, given=g))
    return dict(name=name or 'shape_l%d_p%s' % (sent_len, '-'.join(map(str, predicted))), style='trainmp', X=X, Vde=Vde,
                var_ids=list(predicted), labels=[0] * len(predicted), factors=factors)


def bucket_instances(instances, en_domain, de_domain):
    """Groups parsed instances by sentence shape.  Returns {shape key: dict(spec, var_labels [B][n_pred],
    unary_obs [B][U], rows)} where unary order is the spec's unary factors in id order (= the batched
    engine's unary slots) and `rows` keeps the per-instance records.  Instances without any predicted
    word are dropped (the reference's `initialize` asserts on empty graphs, LBP.py:193-194)."""
    en2id = {w: i for i, w in enumerate(en_domain)}
    de2id = {w: i for i, w in enumerate(de_domain)}
    buckets = {}
    for index, ti in enumerate(instances):
        key, rec = instance_shape(ti, en2id, de2id)
        rec['index'] = index                                             # position in the list handed in (file order)
        if not key[1]:
            continue
        b = buckets.setdefault(key, dict(spec=shape_spec(key[0], key[1], len(en_domain), len(de_domain)), rows=[]))
        b['rows'].append(rec)
    for key, b in buckets.items():
        unary = [f for f in sorted(b['spec']['factors'], key=lambda f: f['id']) if len(f['vars']) == 1]
        labels, obs = [], []
        for r in b['rows']:
            labels.append([r['label'][v] for v in key[1]])
            obs.append([r['de_obs'][f['vars'][0]] if f['factor_type'] == 'en_de' else r['label'][f['given']] for f in unary])
        b['var_labels'] = np.array(labels, dtype=np.int64).reshape(len(b['rows']), len(key[1]))
        b['unary_obs'] = np.array(obs, dtype=np.int64).reshape(len(b['rows']), len(unary))
    return buckets


# ---- params file (train_mp.py:49-102) -----------------------------------------------------------------
def save_params(path, ee_theta, ed_theta, ee_names=EE_NAMES, ed_names=ED_NAMES, d2t=None):
    d2t = d2t or {}

    def row(name, theta):
        return '\t'.join([name.ljust(15)] + ['%0.6f' % v for v in np.asarray(theta, dtype=np.float64).reshape(-1)]) + '\n'
    with codecs.open(path, 'w', 'utf8') as w:
        w.write('\t'.join(['EE_F:'] + list(ee_names)) + '\n')
        w.write(row('Original', ee_theta))
        for (ft, d), t in d2t.items():
            if ft == 'en_en':
                w.write(row(str(d), t))
        w.write('\t'.join(['ED_F:'] + list(ed_names)) + '\n')
        w.write(row('Original', ed_theta))
        for (ft, d), t in d2t.items():
            if ft == 'en_de':
                w.write(row(str(d), t))


def read_params(path):
    """-> (ee_names, ee_theta (1,F), ed_names, ed_theta (1,F), d2t)."""
    text = codecs.open(path, 'r', 'utf8').read()
    p1, p2 = text.split('ED_F:')
    _, p1 = p1.strip().split('EE_F:')

    def block(lines, ft, d2t):
        names = lines[0].split()
        theta = np.array([float(v) for v in lines[1].split()[1:]]).reshape(1, -1)
        for l in lines[2:]:
            items = l.split()
            if items:
                d2t[ft, items[0].strip()] = np.array([float(v) for v in items[1:]]).reshape(1, -1)
        return names, theta
    d2t = {}
    een, eet = block(p1.strip().split('\n'), 'en_en', d2t)
    edn, edt = block(p2.strip().split('\n'), 'en_de', d2t)
    return een, eet, edn, edt, d2t


def prediction_block(sent_id, fg):
    """The text batch_predictions emits per instance (train_mp.py:337): '*SENT_ID:<id>' then the lines
    of FactorGraph.to_string() (LBP.py:109-123) -- what eval.py / get_acc.py parse."""
    return '\n'.join(['*SENT_ID:' + str(sent_id)] + fg.to_string())


def prediction_text(row, predicted, en_domain, top_idx, top_logs, label_logs, log_marginals):
    """The two texts `batch_predictions` returns for one instance (train_mp.py:337-338) from batched read-outs instead of a
    FactorGraph: ('*SENT_ID:' block = FactorGraph.to_string, LBP.py:109-123; .dist lines = to_dist, LBP.py:125-143).
    row: the instance record of bucket_instances; predicted: its predicted positions (= variable order of the batch);
    top_idx / top_logs [n_pred][top]: words of get_max_vocab(50) in descending probability and their log-probabilities;
    label_logs [n_pred]: log p(user's guess); log_marginals [n_pred][X]."""
    lines, dist = ['*SENT_ID:' + str(row['sent_id'])], []
    for pos in range(len(row['label'])):
        if pos in predicted:
            v = predicted.index(pos)
            guess = en_domain[row['label'][pos]]
            pred = ' '.join('%s %0.4f' % (en_domain[int(i)], l) for i, l in zip(top_idx[v], top_logs[v]))
            lines.append(' '.join([row['words'][pos], guess, '%0.4f' % label_logs[v], pred]))
            truth = row['truth'][pos] if row['truth'][pos] is not None else 'None'
            dist.append(' ||| '.join([truth, guess, ' '.join('%0.6f' % x for x in log_marginals[v])]))
        else:
            lines.append(' '.join(['', en_domain[row['label'][pos]], '']))          # a given word: its en_en factors' label
    return '\n'.join(lines), '\n'.join(dist)


# ---- synthetic TI_DIR ---------------------------------------------------------------------------------
def synthesize(directory, n_instances=64, X=64, Vde=64, sent_len=(6, 10), n_predicted=(1, 3), seed=0):
    """Writes a self-contained synthetic TI_DIR: ti (JSON lines), vocab.en, vocab.de and the four
    feature matrices in np.loadtxt form.  Returns a dict of the file paths."""
    rs = np.random.RandomState(seed)
    os.makedirs(directory, exist_ok=True)
    en = ['en%03d' % i for i in range(X)]
    de = ['de%03d' % i for i in range(Vde)]
    paths = dict(ti=os.path.join(directory, 'ti'), end=os.path.join(directory, 'vocab.en'),
                 ded=os.path.join(directory, 'vocab.de'), phi_pmi=os.path.join(directory, 'phi.pmi'),
                 phi_pmi_w1=os.path.join(directory, 'phi.pmi_w1'), phi_ed=os.path.join(directory, 'phi.ed'),
                 phi_ped=os.path.join(directory, 'phi.ped'))
    for p, words in ((paths['end'], en), (paths['ded'], de)):
        with codecs.open(p, 'w', 'utf8') as f:
            f.write('\n'.join(words) + '\n')
    np.savetxt(paths['phi_pmi'], rs.randn(X, X) * 0.5)
    np.savetxt(paths['phi_pmi_w1'], rs.randn(X, X) * 0.5)
    np.savetxt(paths['phi_ed'], -rs.rand(X, Vde))
    np.savetxt(paths['phi_ped'], -rs.rand(X, Vde))
    with codecs.open(paths['ti'], 'w', 'utf8') as f:
        for s in range(n_instances):
            L = int(rs.randint(sent_len[0], sent_len[1] + 1))
            npred = int(rs.randint(n_predicted[0], min(n_predicted[1], L - 1) + 1))
            de_pos = sorted(rs.choice(L, size=min(L, npred + int(rs.randint(0, 2))), replace=False).tolist())
            pred_pos = sorted(rs.choice(de_pos, size=npred, replace=False).tolist())
            sent, guesses, revealed = [], [], []
            for pos in range(L):
                nid = [s, pos]
                if pos in de_pos:
                    truth = en[int(rs.randint(X))]
                    sent.append(dict(sent_id=s, id=nid, l2_word=de[int(rs.randint(Vde))], l1_parent=truth.upper(),
                                     position=pos, lang='de'))
                    guess = truth if rs.rand() < 0.3 else en[int(rs.randint(X))]
                    rec = dict(id=nid, guess=guess, revealed=pos not in pred_pos, l2_word=sent[-1]['l2_word'], reference=truth)
                    (guesses if pos in pred_pos else revealed).append(rec)
                else:
                    sent.append(dict(sent_id=s, id=nid, l2_word=en[int(rs.randint(X))].capitalize(), l1_parent='', position=pos,
                                     lang='en'))
            past = [dict(id=[s, 100 + k], guess=en[int(rs.randint(X))], revealed=False, l2_word=de[int(rs.randint(Vde))],
                         reference=None) for k in range(int(rs.randint(0, 3)))]
            tried = []
            for g in guesses:                  # history that concerns words of THIS sentence (reaches the graph)
                if rs.rand() < 0.5:
                    past.append(dict(id=[s, 200 + len(past)], guess=en[int(rs.randint(X))], revealed=False,
                                     l2_word=g['l2_word'], reference=None))
                if rs.rand() < 0.5:
                    tried.append(dict(id=g['id'], guess=en[int(rs.randint(X))], revealed=bool(rs.rand() < 0.2),
                                      l2_word=g['l2_word'], reference=g['reference']))
            f.write(json.dumps(dict(user_id='u%d' % int(rs.randint(5)), past_correct_guesses=past, past_sentences_seen=[],
                                    past_guesses_for_current_sent=tried, current_sent=sent,
                                    current_revealed_guesses=revealed, current_guesses=guesses)) + '\n')
    return paths
