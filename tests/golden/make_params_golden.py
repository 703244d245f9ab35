#!/usr/bin/env python3
"""Pin the params-file format on the reference's own save_params / read_params.

    python tests/golden/make_params_golden.py [--reference /root/reference]

`train_mp.py` is Python 2 and imports `training_classes` (which needs `enchant`, absent here), so the module cannot be
imported.  Its two params functions (train_mp.py:49-102) use nothing but `codecs` and `numpy`: this script reads the
file as text, passes it through the stdlib `lib2to3` fixers IN MEMORY (as make_golden.py does for LBP.py), takes the
two function definitions out of the resulting syntax tree, and executes THEM -- nothing derived from the reference's text is
written to the repository.  Saved: `params_reference.txt`, the bytes save_params wrote for the seeded inputs below (an
output file of the reference), and `params_reference.npz`, the inputs and what read_params returned for that file.
Needs /root/reference; never run on the GPU box."""
import argparse
import ast
import codecs
import hashlib
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_params_functions(ref_root):
    from lib2to3 import refactor
    path = os.path.join(ref_root, 'train_mp.py')
    text = open(path).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    py3 = str(tool.refactor_string(text + '\n', 'train_mp.py'))
    tree = ast.parse(py3)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('read_params', 'save_params')]
    assert sorted(n.name for n in wanted) == ['read_params', 'save_params']
    ns = {'np': np, 'codecs': codecs}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), path, 'exec'), ns)
    return ns['read_params'], ns['save_params'], hashlib.sha256(open(path, 'rb').read()).hexdigest()


def inputs():
    rs = np.random.RandomState(20260)
    ee_names = ['pmi', 'pmi_w1', 'bias']
    ed_names = ['ed', 'ped', 'length', 'correct', 'history', 'session_history']
    ee = rs.randn(1, 3) * 3
    ed = rs.randn(1, 6) * 3
    ee[0, 1] = 0.0000004            # rounds to 0.000000
    ed[0, 2] = -12345.6789012
    d2t = {}
    for d in ['u17', 'a_rather_long_user_name_x', 'ünï']:        # a name longer than the 15-column pad; non-ASCII
        d2t['en_en', d] = rs.randn(1, 3)
        d2t['en_de', d] = rs.randn(1, 6)
    return ee_names, ee, ed_names, ed, d2t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    a = ap.parse_args()
    read_params, save_params, sha = load_params_functions(a.reference)
    ee_names, ee, ed_names, ed, d2t = inputs()
    out = os.path.join(HERE, 'params_reference.txt')
    save_params(codecs.open(out, 'w', 'utf8'), ee, ed, ee_names, ed_names, d2t)
    een, eet, edn, edt, got = read_params(out)
    keys = sorted(got)
    np.savez(os.path.join(HERE, 'params_reference.npz'), ee=ee, ed=ed, eet=eet, edt=edt,
             een=np.array(een), edn=np.array(edn),
             in_keys=np.array(['%s\t%s' % k for k in d2t]), in_vals_ee=np.stack([d2t[k] for k in d2t if k[0] == 'en_en']),
             in_vals_ed=np.stack([d2t[k] for k in d2t if k[0] == 'en_de']),
             out_keys=np.array(['%s\t%s' % k for k in keys]),
             out_vals_ee=np.stack([got[k] for k in keys if k[0] == 'en_en']), out_vals_ed=np.stack([got[k] for k in keys if k[0] == 'en_de']))
    man_path = os.path.join(HERE, 'MANIFEST.json')
    man = json.load(open(man_path)) if os.path.exists(man_path) else {}
    man['params_reference'] = {'reference_file': 'train_mp.py', 'sha256': sha, 'functions': ['read_params', 'save_params'],
                               'generator': 'tests/golden/make_params_golden.py'}
    json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
    print('wrote', out, 'and params_reference.npz;', len(got), 'adapted rows read back')


if __name__ == '__main__':
    main()
