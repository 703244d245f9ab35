#!/usr/bin/env python3
"""Pin the prediction / .dist text formats on the reference's own FactorGraph.to_string /
to_dist (LBP.py:109-143) and the '*SENT_ID:' block of train_mp.py:337.

    python tests/golden/make_text_golden.py [--reference /root/reference]

Runs the reference exactly as make_golden.py does (LBP.py through lib2to3 in memory, c_array_utils cythonized in a
temporary directory, plus an ordering for FactorNode, which Python 2 had implicitly) on the user_k3_x64 case with word labels set, three sweeps with the case's roots, and saves the
TEXT the reference emits -- an output of the reference -- as user_k3_x64_text.json.  Needs /root/reference."""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402
import make_golden as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    a = ap.parse_args()
    L, au, cleanup = G.load_reference(a.reference)
    try:
        roots = G.Roots(L)
        # Python 2 orders any two objects; `sorted([(f.position, f) ...])` (LBP.py:110, 127) leans on that when several
        # factors share a position (train_mp.py:295: one en_en factor per predicted variable at a given word's position).
        # The tied factors carry the same type and word label, so the text does not depend on their order: give the
        # class an order for the run.
        L.FactorNode.__lt__ = lambda self, other: self.id < other.id
        case = [c for c in C.inference_cases() if c['name'] == 'user_k3_x64'][0]
        spec = case['spec']
        fg = C.build_graph(L, spec, C.make_inputs(spec, case['seed']))
        for f in fg.factors:
            f.word_label = 'w%d' % (f.position or 0)
        roots.queue = [case['roots'][0]]
        fg.initialize()
        roots.queue = list(case['roots'][:3])
        fg.treelike_inference(3)
        strings = fg.to_string()
        out = {'case': case['name'], 'roots': list(case['roots'][:3]), 'to_string': strings, 'to_dist': fg.to_dist(),
               'prediction_block': '\n'.join(['*SENT_ID:' + str(17)] + strings)}
        json.dump(out, open(os.path.join(HERE, 'user_k3_x64_text.json'), 'w'), indent=1, ensure_ascii=False)
        print('wrote user_k3_x64_text.json:', len(strings), 'to_string lines,', len(out['to_dist'].split('\n')), 'to_dist lines')
    finally:
        cleanup()


if __name__ == '__main__':
    main()
