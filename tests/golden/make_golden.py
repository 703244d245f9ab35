#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING the reference itself.

    python tests/golden/make_golden.py [--reference /root/reference] [--only NAME]

The reference is Python 2 + one Cython module and cannot be imported as-is under Python 3.10 /
NumPy 2.  This script therefore applies the survey's mechanical recipe at
run time, never writing anything derived from the reference into the repository:

  * `LBP.py` is read as text, passed through the stdlib `lib2to3` fixers IN MEMORY (touches only
    `print`, `.iteritems()`, `dict.keys()/values()`), and exec'd as a module object;
  * `array_utils/c_array_utils.pyx` gets three dtype-token substitutions (`np.int_t`->`np.int64_t`,
    `dtype=np.int`->`np.int64`, `dtype=np.float`->`np.float64`; the tokens no longer exist in
    NumPy 2) and is cythonized with language_level=2 inside a `tempfile.mkdtemp()` directory that
    is deleted before the script exits.

Only numeric outputs (and the seeds/specs that produced them, see cases.py) are saved as `.npz`.
The sha256 of both reference files is recorded in MANIFEST.json so a fixture can be tied to the
exact reference revision.  This script needs /root/reference and is never run on the GPU box.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402


def sha256(path):
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def load_reference(ref_root):
    """Returns (LBP module, au module, cleanup fn)."""
    from lib2to3 import refactor
    tmp = tempfile.mkdtemp(prefix='mlbp_ref_')
    # --- Cython module ---------------------------------------------------------------------------
    pyx = open(os.path.join(ref_root, 'array_utils', 'c_array_utils.pyx')).read()
    pyx = pyx.replace('np.int_t', 'np.int64_t')
    pyx = re.sub(r'dtype=np\.int\b', 'dtype=np.int64', pyx)
    pyx = re.sub(r'dtype=np\.float\b', 'dtype=np.float64', pyx)
    with open(os.path.join(tmp, 'c_array_utils.pyx'), 'w') as f:
        f.write(pyx)
    with open(os.path.join(tmp, 'setup.py'), 'w') as f:
        f.write("from setuptools import setup\nfrom Cython.Build import cythonize\nimport numpy\n"
                "setup(ext_modules=cythonize('c_array_utils.pyx', language_level=2),"
                " include_dirs=[numpy.get_include()])\n")
    subprocess.check_call([sys.executable, 'setup.py', '-q', 'build_ext', '--inplace'], cwd=tmp,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = [p for p in os.listdir(tmp) if p.startswith('c_array_utils') and p.endswith('.so')][0]
    spec = importlib.util.spec_from_file_location('c_array_utils', os.path.join(tmp, so))
    au = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(au)
    pkg = types.ModuleType('array_utils')
    pkg.c_array_utils = au
    sys.modules['array_utils'] = pkg
    sys.modules['array_utils.c_array_utils'] = au
    # --- LBP.py, in memory ------------------------------------------------------------------------
    src = open(os.path.join(ref_root, 'LBP.py')).read()
    if not src.endswith('\n'):
        src += '\n'
    rt = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    py3 = str(rt.refactor_string(src, 'LBP.py'))
    L = types.ModuleType('LBP_reference')
    exec(compile(py3, '<reference LBP.py via lib2to3>', 'exec'), L.__dict__)

    def cleanup():
        shutil.rmtree(tmp, ignore_errors=True)
    return L, au, cleanup


class Roots:
    """Replaces `random.sample` inside the reference module so roots are inputs, not RNG draws
    (LBP.py:176, 223)."""

    def __init__(self, L):
        self.queue = []
        L.random = types.SimpleNamespace(sample=self.sample)

    def sample(self, population, k):
        assert k == 1
        r = self.queue.pop(0)
        assert r in population, (r, population)
        return [r]


def stack_msgs(fg, keys):
    return np.stack([fg.messages[k].m.reshape(-1) for k in keys])


def encode_schedule(sched):
    return np.array([[*C.node_code(str(a)), *C.node_code(str(b))] for a, b in sched], dtype=np.int64)


def run_inference_case(L, roots, case, approx=False):
    spec = case['spec']
    inputs = C.make_inputs(spec, case['seed'], case['kind'] or 'uniform')
    fg = C.build_graph(L, spec, inputs)
    fg.learning_rate = 0.05
    fg.regularization_param = 0.2 / 17.0
    if approx:
        fg.use_approx_inference = True
        fg.use_approx_beliefs = True
    keys = C.msg_keys(spec)
    out = {}
    roots.queue = [case['roots'][0]]
    fg.initialize()
    out['is_loopy'] = np.array(bool(fg.isLoopy))
    assert sorted(fg.messages.keys()) == sorted(keys)
    out['msgs_init'] = stack_msgs(fg, keys)
    if case['force_loopy']:
        fg.isLoopy = True
    if 'request' in case:                      # one call, reference decides how many sweeps run
        roots.queue = list(case['roots'])
        fg.treelike_inference(case['request'])
        out['msgs_s%d' % case['snaps'][0]] = stack_msgs(fg, keys)
        out['roots_consumed'] = np.array(len(case['roots']) - len(roots.queue))
    else:
        assert fg.isLoopy
        done = 0
        for s in case['snaps']:
            while done < s:
                roots.queue = [case['roots'][done]]
                fg.treelike_inference(1)
                done += 1
            out['msgs_s%d' % s] = stack_msgs(fg, keys)
    for r in sorted(set(case['roots'])):
        out['sched_root%d' % r] = encode_schedule(fg.get_message_schedule(fg.variables[r]))
    # ---- read-outs ----
    vorder = list(fg.variables.keys())
    out['var_order'] = np.array(vorder, dtype=np.int64)
    out['marginals'] = np.stack([fg.variables[v].get_marginal().m.reshape(-1) for v in vorder])
    out['log_posterior'] = np.array(fg.get_posterior_probs())
    X = spec['X']
    if X >= 50:
        top = []
        for v in vorder:
            sl, slp, pred = fg.variables[v].get_max_vocab(50)
            top.append([int(w[1:]) for w, _ in pred])
        out['top50'] = np.array(top, dtype=np.int64)
        out['precision_counts'] = np.array(fg.get_precision_counts(), dtype=np.int64)
    if not case.get('light'):
        for f in fg.factors:
            b = f.get_factor_beliefs()
            out['belief_F%d' % f.id] = np.array(b)
        if spec['style'] == 'trainmp':
            for f in fg.factors:
                out['grad_F%d' % f.id] = f.get_gradient()
            g_ee, g_ed = fg.get_unregularized_gradeint()
            out['grad_unreg_en_en'], out['grad_unreg_en_de'] = g_ee, g_ed
            g_ed2, g_ee2 = fg.get_gradient()
            out['grad_reg_en_en'], out['grad_reg_en_de'] = g_ee2, g_ed2
            r_ee, r_ed = fg.return_gradient()
            out['grad_ret_en_en'], out['grad_ret_en_de'] = r_ee, r_ed
    return out


def run_schedule_cases(L, roots):
    out = {}
    for spec in C.schedule_topologies():
        inputs = C.make_inputs(spec, 1, 'uniform')
        fg = C.build_graph(L, spec, inputs)
        for vid in spec['var_ids']:
            if vid not in fg.variables:
                continue
            roots.queue = [vid]
            out['%s/loops_root%d' % (spec['name'], vid)] = np.array(bool(fg.has_loops()))
            out['%s/sched_root%d' % (spec['name'], vid)] = encode_schedule(
                fg.get_message_schedule(fg.variables[vid]))
    return out


def au_inputs(X, seed):
    rs = np.random.RandomState(seed)
    return dict(m1=rs.rand(X, 1), m2=rs.rand(X, 1), T=rs.rand(X, X) + 0.01, T2=np.exp(rs.randn(X, X)),
                c=rs.rand(X, 1) ** 4, r=rs.rand(1, X) ** 4)


AU_SIZES = (('x4', 4, 4004), ('x64', 64, 4064), ('x128', 128, 4128), ('x128b', 128, 5128))


def exc_record(fn):
    try:
        fn()
    except BaseException as e:  # noqa: B902  (the reference raises BaseException in places)
        return '%s: %s' % (type(e).__name__, e)
    return 'no exception'


def run_au_cases(au):
    out, errs = {}, {}
    for tag, X, seed in AU_SIZES:
        i = au_inputs(X, seed)
        p = tag + '/'
        out[p + 'pointwise_multiply'] = au.pointwise_multiply(i['m1'], i['m2'])
        out[p + 'dense_pointwise_multiply'] = au.dense_pointwise_multiply(i['T'], i['T2'])
        out[p + 'normalize_vec'] = au.normalize(i['m1'].copy())
        out[p + 'normalize_mat'] = au.normalize(i['T'].copy())
        z = np.zeros((X, 1))
        zr = au.normalize(z)
        out[p + 'normalize_zero_is_same_object'] = np.array(zr is z)
        neg = -i['m1']
        out[p + 'normalize_negative_sum'] = au.normalize(neg)
        out[p + 'normalize_negative_sum_inplace'] = np.array(neg)
        out[p + 'dense_dot_Tm'] = au.dense_dot(i['T'], i['m1'])
        out[p + 'dense_dot_mT'] = au.dense_dot(i['m1'].T, i['T'])
        out[p + 'dense_dot_outer'] = au.dense_dot(i['c'], i['r'])
        if X >= 100:
            out[p + 'sparse_vec_mat_dot_col'] = au.sparse_vec_mat_dot(i['c'], i['T'])
            out[p + 'sparse_vec_mat_dot_row'] = au.sparse_vec_mat_dot(i['r'], i['T'])
            sp, ci, ri = au.sparse_dot(i['c'], i['r'])
            out[p + 'sparse_dot'] = sp
            out[p + 'sparse_dot_cidx_sorted'] = np.sort(ci).astype(np.int64)
            out[p + 'sparse_dot_ridx_sorted'] = np.sort(ri).astype(np.int64)
            spm = au.sparse_pointwise_multiply(sp, ci, ri, i['T'])
            out[p + 'sparse_pointwise_multiply'] = spm.copy()
            spn = au.sparse_normalize(spm, ci, ri)
            out[p + 'sparse_normalize'] = spn
            out[p + 'sparse_normalize_is_same_object'] = np.array(spn is spm)
        else:
            errs[p + 'sparse_vec_mat_dot_col'] = exc_record(lambda: au.sparse_vec_mat_dot(i['c'], i['T']))
            errs[p + 'sparse_dot'] = exc_record(lambda: au.sparse_dot(i['c'], i['r']))
    i = au_inputs(8, 1)
    errs['dense_dot_float32'] = exc_record(lambda: au.dense_dot(i['T'].astype(np.float32), i['m1']))
    errs['dense_dot_ndim1'] = exc_record(lambda: au.dense_dot(i['T'], i['m1'].reshape(-1)))
    errs['dense_pointwise_multiply_ndim1'] = exc_record(
        lambda: au.dense_pointwise_multiply(i['m1'].reshape(-1), i['m1'].reshape(-1)))
    errs['dir'] = sorted(n for n in dir(au) if not n.startswith('_'))
    return out, errs


class _Py2Dict(dict):
    """A dict the way the reference's Python 2 code addresses one (`.iteritems()`): an input adapter, the reference's
    function runs unmodified."""
    iteritems = dict.items


def run_au_dormant_cases(au):
    """The 12 functions of c_array_utils.pyx that nothing in the reference calls (pyx:18-20, 43-75, 96-105, 132-190), run on
    seeded inputs so that the drop-in's bodies are pinned like the live ones.  Index SETS are stored sorted (np.argpartition
    leaves the order open); dicts as sorted key arrays + values."""
    from scipy import sparse
    out, errs = {}, {}
    rs = np.random.RandomState(9001)
    # clip: in place, returns its argument
    m = rs.rand(6, 5) * np.where(rs.rand(6, 5) < 0.4, 1e-120, 1.0)
    out['clip/in'] = m.copy()
    r = au.clip(m)
    out['clip/out'], out['clip/same_object'] = r.copy(), np.array(r is m)
    # induce_s_pointwise_multiply_clip: the K = 100 largest cells of d1, times d2
    d1, d2 = rs.rand(16, 16), rs.rand(16, 16) + 0.5
    out['ispmc/d1'], out['ispmc/d2'], out['ispmc/out'] = d1, d2, au.induce_s_pointwise_multiply_clip(d1, d2)
    errs['ispmc/size_100'] = exc_record(lambda: au.induce_s_pointwise_multiply_clip(rs.rand(10, 10), rs.rand(10, 10)))
    # induce_s: column vectors of 64 (returned as it is), 128 (top 100 kept) and exactly 100 entries
    for X in (64, 128):
        v = rs.rand(X, 1)
        r = au.induce_s(v)
        out['induce_s/x%d/in' % X], out['induce_s/x%d/out' % X], out['induce_s/x%d/same_object' % X] = v, np.array(r), np.array(r is v)
    errs['induce_s/x100'] = exc_record(lambda: au.induce_s(rs.rand(100, 1)))
    # induce_s_mutliply_clip: d2 (a x b, a < b) . the K entries of s1 (b x 1) that are largest in magnitude
    s1, dd = rs.randn(128, 1), rs.rand(64, 128)
    out['ismc/s1'], out['ismc/d2'], out['ismc/out'] = s1, dd, au.induce_s_mutliply_clip(s1, dd)
    errs['ismc/b_99'] = exc_record(lambda: au.induce_s_mutliply_clip(rs.randn(99, 1), rs.rand(64, 99)))
    # make_sparse_and_dot: {(x, y): m1[x, 0] * m2[0, y]} over the top-K x top-K index pairs
    m1, m2 = rs.rand(128, 1), rs.rand(1, 128)
    d = au.make_sparse_and_dot(m1, m2)
    keys = np.array(sorted((int(x), int(y)) for x, y in d), dtype=np.int64)
    out['msad/m1'], out['msad/m2'], out['msad/keys'] = m1, m2, keys
    out['msad/values'] = np.array([d[x, y] for x, y in keys])
    # sparse_multiply_and_normalize: dict of cells -> (dense array, dict), both normalised over the cells
    cells = _Py2Dict({(int(x), int(y)): float(v) for x, y, v in zip(rs.randint(0, 12, 20), rs.randint(0, 9, 20), rs.rand(20))})
    mm = rs.rand(12, 9) + 0.1
    z, zd = au.sparse_multiply_and_normalize(cells, mm)
    ck = np.array(sorted(cells), dtype=np.int64)
    out['smn/cell_keys'], out['smn/cell_values'], out['smn/m2'] = ck, np.array([cells[x, y] for x, y in ck]), mm
    out['smn/dense'], out['smn/dict_values'] = z, np.array([zd[x, y] for x, y in ck])
    # sd_matrix_multiply / ss_matix_multiply: `.dot` of a scipy sparse matrix with a dense resp. sparse one
    sa = sparse.random(9, 14, density=0.3, random_state=rs, format='csr')
    sb = sparse.random(14, 6, density=0.4, random_state=rs, format='csr')
    db = rs.rand(14, 5)
    out['sdmm/a'], out['sdmm/b'], out['sdmm/out'] = sa.toarray(), db, np.asarray(au.sd_matrix_multiply(sa, db))
    r = au.ss_matix_multiply(sa, sb)
    out['ssmm/a'], out['ssmm/b'], out['ssmm/out'], out['ssmm/out_is_sparse'] = sa.toarray(), sb.toarray(), r.toarray(), np.array(sparse.issparse(r))
    # the adaptation helpers
    phi = rs.rand(7, 4)
    ap = au.make_adapt_phi(phi, 3)
    out['adapt/phi'], out['adapt/make'] = phi, ap.copy()
    r = au.set_adaptation(4, ap, [1, 3])
    out['adapt/set'], out['adapt/set_same_object'] = r.copy(), np.array(r is ap)
    r = au.set_adaptation_off(4, ap, [3])
    out['adapt/off'], out['adapt/off_same_object'] = r.copy(), np.array(r is ap)
    phi2 = rs.rand(7, 4)
    r = au.set_original(phi2, ap)
    out['adapt/phi2'], out['adapt/original'], out['adapt/original_same_object'] = phi2, r.copy(), np.array(r is ap)
    errs['induce_s_multiply_threshold'] = exc_record(lambda: au.induce_s_multiply_threshold(s1, dd))
    errs['sd_pointwise_multiply'] = exc_record(lambda: au.sd_pointwise_multiply(sa, db))
    return out, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    L, au, cleanup = load_reference(a.reference)
    try:
        roots = Roots(L)
        manifest = dict(reference_sha256={
            'LBP.py': sha256(os.path.join(a.reference, 'LBP.py')),
            'array_utils/c_array_utils.pyx': sha256(os.path.join(a.reference, 'array_utils', 'c_array_utils.pyx'))},
            numpy=np.__version__, python=sys.version.split()[0], files={})

        def save(name, arrays):
            if a.only and a.only != name:
                return
            path = os.path.join(HERE, name + '.npz')
            np.savez_compressed(path, **arrays)
            manifest['files'][name + '.npz'] = sorted(arrays.keys())
            print('%-28s %3d arrays %8.1f KiB' % (name, len(arrays), os.path.getsize(path) / 1024.0))

        for case in C.inference_cases():
            save(case['name'], run_inference_case(L, roots, case))
        for case in C.approx_cases():
            save(case['name'], run_inference_case(L, roots, case, approx=True))
        save('schedules', run_schedule_cases(L, roots))
        au_out, au_err = run_au_cases(au)
        save('au_functions', au_out)
        manifest['au_errors'] = au_err
        dorm_out, dorm_err = run_au_dormant_cases(au)
        save('au_dormant_functions', dorm_out)
        manifest['au_dormant_errors'] = dorm_err
        if a.only == 'au_dormant_functions':      # added in round 4: the other fixtures stay as they are, the manifest gains two keys
            old = json.load(open(os.path.join(HERE, 'MANIFEST.json')))
            old['files'].update(manifest['files'])
            old['au_dormant_errors'] = dorm_err
            with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
                json.dump(old, f, indent=1, sort_keys=True)
        if not a.only:
            with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
                json.dump(manifest, f, indent=1, sort_keys=True)
    finally:
        cleanup()


if __name__ == '__main__':
    main()
