"""Data formats around the hot path (macaronicusermodeling_amd.tidir): TI JSON schema and
normalisation rules, vocab / feature files, the instance -> shape compiler, params file round trip.
Host-side only (no GPU)."""
import json
import os

import numpy as np
import pytest

import cases as C
from macaronicusermodeling_amd import tidir


def test_guess_and_node_normalisation_rules():
    # training_classes.py:97-110
    assert tidir.normalize_guess('  ') == '__blank__'
    assert tidir.normalize_guess('__UNK__') == '__UNK__'              # kept verbatim
    assert tidir.normalize_guess("Don't") == 'dont'
    assert tidir.normalize_guess('the quick fox') == 'quick'          # longest token of a phrasal guess
    assert tidir.normalize_guess('word*') == 'word'
    assert tidir.normalize_guess('*') == '*'
    n = tidir.parse_node(dict(sent_id=3, id=[3, 1], l2_word="It's", l1_parent='X', position='2', lang='en'))
    assert n['l2_word'] == 'its' and n['position'] == 2 and n['id'] == (3, 1) and n['l1_parent'] == 'X'
    n = tidir.parse_node(dict(sent_id=3, id=[3, 2], l2_word='Haus', l1_parent="House's", position=0, lang='de'))
    assert n['l2_word'] == 'Haus' and n['l1_parent'] == 'houses'


def test_shape_spec_equals_the_case_generator():
    """Two independent restatements of train_mp.py:257-299 must agree on ids, order, gaps."""
    for L, pred in [(10, [1, 4, 7]), (6, [0, 3]), (5, [2]), (9, [0, 1, 3, 5, 7])]:
        a = tidir.shape_spec(L, pred, 16, 12)
        b = C.user_spec(L, pred, 16, 12)
        assert a['var_ids'] == b['var_ids']
        for fa, fb in zip(a['factors'], b['factors']):
            for k in ('id', 'vars', 'dims', 'factor_type', 'gap', 'obs_size', 'position'):
                assert fa[k] == fb[k], (k, fa, fb)
        assert len(a['factors']) == len(b['factors'])


def test_synthetic_ti_dir_round_trip(tmp_path):
    paths = tidir.synthesize(str(tmp_path), n_instances=40, X=16, Vde=12, seed=3)
    en, de = tidir.read_vocab(paths['end']), tidir.read_vocab(paths['ded'])
    assert len(en) == 16 and len(de) == 12
    phi_ee, phi_w1, phi_ed = tidir.load_features(paths['phi_pmi'], paths['phi_pmi_w1'], paths['phi_ed'], paths['phi_ped'])
    assert phi_ee.shape == (16, 16, 3) and phi_w1.shape == (16, 16, 3) and phi_ed.shape == (16, 12, 6)
    assert np.all(phi_ee[:, :, 1] == 0) and np.all(phi_ee[:, :, 2] == 1)          # train_mp.py:602-603
    assert np.all(phi_ed[:, :, 2:5] == 0) and np.all(phi_ed[:, :, 5] == 1)        # train_mp.py:614-619
    inst = tidir.read_instances(paths['ti'])
    assert len(inst) == 40
    buckets = tidir.bucket_instances(inst, en, de)
    assert sum(len(b['rows']) for b in buckets.values()) == 40
    for (L, pred), b in buckets.items():
        spec = b['spec']
        n_given = L - len(pred)
        U = len(pred) * (1 + n_given)
        assert b['var_labels'].shape == (len(b['rows']), len(pred))
        assert b['unary_obs'].shape == (len(b['rows']), U)
        assert len([f for f in spec['factors'] if len(f['vars']) == 2]) == len(pred) * (len(pred) - 1) // 2
        # observed columns: en_de factors index the de vocabulary, en_en ones the en vocabulary
        unary = [f for f in spec['factors'] if len(f['vars']) == 1]
        for u, f in enumerate(unary):
            lim = 12 if f['factor_type'] == 'en_de' else 16
            assert b['unary_obs'][:, u].min() >= 0 and b['unary_obs'][:, u].max() < lim
    # one instance by hand
    raw = json.loads(open(paths['ti']).readline())
    key, rec = tidir.instance_shape(tidir.parse_instance(raw), {w: i for i, w in enumerate(en)},
                                    {w: i for i, w in enumerate(de)})
    guessed = sorted(n['position'] for n in raw['current_sent']
                     if any(g['id'] == n['id'] for g in raw['current_guesses']))
    assert list(key[1]) == guessed and key[0] == len(raw['current_sent'])


def test_params_file_round_trip(tmp_path):
    p = os.path.join(str(tmp_path), 'params')
    ee = np.array([[0.125, -1.5, 2.0]])
    ed = np.array([[0.1, 0.2, 0.3, -0.4, 0.5, 0.000001]])
    d2t = {('en_en', 'u1'): ee * 2, ('en_de', 'u1'): ed * 3}
    tidir.save_params(p, ee, ed, d2t=d2t)
    lines = open(p).read().split('\n')
    assert lines[0] == 'EE_F:\tpmi\tpmi_w1\tbias'                                     # train_mp.py:81
    assert lines[1] == 'Original'.ljust(15) + '\t0.125000\t-1.500000\t2.000000'        # train_mp.py:82-84
    een, eet, edn, edt, got = tidir.read_params(p)
    assert een == tidir.EE_NAMES and edn == tidir.ED_NAMES
    np.testing.assert_allclose(eet, ee, atol=1e-6); np.testing.assert_allclose(edt, ed, atol=1e-6)
    np.testing.assert_allclose(got['en_en', 'u1'], ee * 2, atol=1e-6)
    np.testing.assert_allclose(got['en_de', 'u1'], ed * 3, atol=1e-6)


def test_params_file_matches_the_reference_functions(tmp_path):
    """SURVEY 8 row f4, pinned: tests/golden/params_reference.txt is what the reference's OWN save_params
    (train_mp.py:80-102) wrote for seeded thetas and three adapted domains (one name longer than the 15-column pad, one
    non-ASCII), params_reference.npz what its read_params (train_mp.py:49-77) read back (make_params_golden.py runs the
    two functions out of the reference's file).  Our writer must produce the same bytes, our reader the same arrays."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ref_bytes = open(os.path.join(here, 'params_reference.txt'), 'rb').read()
    z = np.load(os.path.join(here, 'params_reference.npz'))
    keys = [tuple(k.split('\t')) for k in z['in_keys']]
    ee_rows = iter(z['in_vals_ee']); ed_rows = iter(z['in_vals_ed'])
    d2t = {}
    for k in keys:                                          # the insertion order the reference iterated in
        d2t[k] = next(ee_rows) if k[0] == 'en_en' else next(ed_rows)
    p = os.path.join(str(tmp_path), 'params')
    tidir.save_params(p, z['ee'], z['ed'], ee_names=list(z['een']), ed_names=list(z['edn']), d2t=d2t)
    assert open(p, 'rb').read() == ref_bytes
    een, eet, edn, edt, got = tidir.read_params(os.path.join(here, 'params_reference.txt'))
    assert een == list(z['een']) and edn == list(z['edn'])
    assert np.array_equal(eet, z['eet']) and np.array_equal(edt, z['edt'])
    out_keys = [tuple(k.split('\t')) for k in z['out_keys']]
    assert sorted(got) == out_keys
    ee_rows = iter(z['out_vals_ee']); ed_rows = iter(z['out_vals_ed'])
    for k in out_keys:
        want = next(ee_rows) if k[0] == 'en_en' else next(ed_rows)
        assert got[k].shape == (1, want.shape[-1]) and np.array_equal(got[k], want)


def _tidir_gold():
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    return json.load(open(os.path.join(here, 'tidir_reference.json'), encoding='utf8'))


def test_normalisation_rules_equal_the_reference_classes():
    """SURVEY 8 row f3, pinned: Guess.__init__ / SimpleNode.__init__ of training_classes.py, run by
    make_tidir_golden.py on a list of spellings (blank, reserved words in any case, phrasal guesses, trailing '*',
    apostrophes, non-ASCII)."""
    g = _tidir_gold()['normalisation']
    for raw, want in g['guesses']:
        assert tidir.normalize_guess(raw) == want, repr(raw)
    for d, (sent_id, nid, l2, l1, pos, lang) in g['nodes']:
        n = tidir.parse_node(d)
        assert (n['sent_id'], list(n['id']), n['l2_word'], n['l1_parent'], n['position'], n['lang']) == (sent_id, nid, l2, l1, pos, lang)


def test_reader_and_shape_compiler_equal_the_reference_graph_builder():
    """SURVEY 8 row f3, pinned: for the 12 instances of tests/golden/tidir_reference.json the reference's own
    TrainingInstance.from_dict and create_factor_graph (train_mp.py:105-306, run by make_tidir_golden.py against the
    reference's LBP.py) give the normalised guesses and nodes, the variables that enter the graph and the factor list in
    creation order; parse_instance / instance_shape / shape_spec / bucket_instances must give the same."""
    gold = _tidir_gold()
    en, de = gold['vocab_en'], gold['vocab_de']
    en2id = {w: i for i, w in enumerate(en)}
    de2id = {w: i for i, w in enumerate(de)}
    n_checked = 0
    for line, ref in zip(gold['instances'], gold['reference']):
        ti = tidir.parse_instance(line)
        for fld, want in ref['guesses'].items():
            got = [[list(g['id']), g['guess'], bool(g['revealed']), g['l2_word'], g['reference']] for g in ti[fld]]
            assert got == want, fld
        assert [[n['sent_id'], list(n['id']), n['l2_word'], n['l1_parent'], n['position'], n['lang']] for n in ti['current_sent']] == ref['nodes']
        key, rec = tidir.instance_shape(ti, en2id, de2id)
        sent = sorted(ti['current_sent'], key=lambda n: n['position'])
        # variables of the graph = the predicted words, id = index in position order (train_mp.py:108-133)
        assert [v[0] for v in ref['variables']] == list(key[1])
        for vid, var_type, label, truth in ref['variables']:
            assert var_type == 'var_type_predicted' and en[rec['label'][vid]] == label and sent[vid]['l1_parent'] == truth
        b = tidir.bucket_instances([ti], en, de)[key]
        spec = b['spec']
        unary = [f for f in sorted(spec['factors'], key=lambda f: f['id']) if len(f['vars']) == 1]
        obs = {f['id']: int(b['unary_obs'][0][u]) for u, f in enumerate(unary)}
        assert len(spec['factors']) == len(ref['factors'])
        for f, (fid, ftype, fvars, observed, gap, position, word_label) in zip(sorted(spec['factors'], key=lambda f: f['id']), ref['factors']):
            assert (f['id'], f['factor_type'], f['vars'], f['gap'], f['position']) == (fid, ftype, fvars, gap, position)
            assert (obs[fid] if len(fvars) == 1 else None) == observed
            if ftype == 'en_de':
                assert de[obs[fid]] == word_label
            elif len(fvars) == 1:
                assert en[obs[fid]] == word_label
            n_checked += 1
    assert n_checked > 60


def test_prediction_writer_reproduces_the_reference_text_from_its_own_distributions():
    """tidir.prediction_text -- what TiDirTrainer.predict(save_predictions=...) writes per instance -- fed with the distributions
    the REFERENCE printed: tests/golden/tidir_batch_reference.json holds, per instance, the '*SENT_ID:' block and the .dist lines
    its batch_predictions returned (train_mp.py:310-343, LBP.py:109-143).  The .dist lines carry every variable's log-marginals
    (6 decimals); parsed and handed back, the writer must give the same .dist text character for character, and the same block:
    same words in the same places and order (given words, guesses, the 50 most probable words by descending probability),
    numbers to the 4 decimals printed (one unit of slack: they are re-rounded from 6-decimal values)."""
    import json
    from macaronicusermodeling_amd import tidir
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tidir_batch_reference.json'), encoding='utf8'))
    en, de = gold['vocab_en'], gold['vocab_de']
    instances = [tidir.parse_instance(l) for l in gold['instances']]
    buckets = tidir.bucket_instances(instances, en, de)
    seen = 0
    for key, b in buckets.items():
        predicted = list(key[1])
        for row in b['rows']:
            ref = gold['predictions'][row['index']]
            logm = np.array([[float(x) for x in line.split(' ||| ')[2].split()] for line in ref['dist'].split('\n')])
            assert logm.shape == (len(predicted), len(en))
            top = np.argsort(-logm, axis=1, kind='stable')[:, :50]
            top_logs = np.take_along_axis(logm, top, axis=1)
            label_logs = [logm[v, row['label'][pos]] for v, pos in enumerate(predicted)]
            block, dist = tidir.prediction_text(row, predicted, en, top, top_logs, label_logs, logm)
            assert dist == ref['dist']
            got, want = block.split('\n'), ref['block'].split('\n')
            assert got[0] == want[0] and len(got) == len(want)
            for gl, wl in zip(got[1:], want[1:]):
                gt, wt = gl.split(' '), wl.split(' ')
                assert len(gt) == len(wt)
                for a_, b_ in zip(gt, wt):
                    try:
                        fa, fb = float(a_), float(b_)
                    except ValueError:
                        assert a_ == b_                      # a word: the same one in the same place
                    else:
                        assert abs(fa - fb) <= 1.0001e-4
            seen += 1
    assert seen == len(gold['predictions'])
