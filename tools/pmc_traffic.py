#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md's HBM
section prescribes) into HBM bytes per launch for the sweep kernels of one bench.py workload.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <batch> <sweeps> <out.json> [key]
(also merges the total into profiles/pmc_traffic.json under `key`, with the sha of the kernel sources)
Rule (gfx950): FETCH_SIZE counts KiB and reads exactly half of a 16-byte-per-lane coalesced stream -> doubled;
WRITE_SIZE (KiB) is taken as is."""
import csv
import json
import os
import sys
from collections import defaultdict


KERNELS = ['sweep_x64_lean_kernel', 'sweep_x64_sf_kernel', 'sweep_x64_fused_kernel', 'sweep_x64_shared_kernel', 'sweep_wide_kernel', 'sweep_generic_kernel',
           'sweep_x64_kernel', 'unary_writeback_kernel', 'table_fragments_kernel']


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        full = r['Kernel_Name']
        hit = [k for k in KERNELS if k in full]
        if not hit:
            continue
        name = hit[0] + (full[full.index(hit[0]) + len(hit[0]):].split('(')[0] if '<' in full else '')
        acc[(name, int(r['Dispatch_Id']))].append(float(r['Counter_Value']))
    out = defaultdict(list)
    for (name, _), vals in acc.items():
        out[name].append(sum(vals))
    return out


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
workload, batch, sweeps, out_path = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
res, total = {}, 0.0
for name in sorted(set(fetch) | set(write)):
    f = fetch.get(name, [0.0])[1:] or fetch.get(name, [0.0])           # drop the first (cold) dispatch when possible
    w = write.get(name, [0.0])[1:] or write.get(name, [0.0])
    mf, mw = sum(f) / len(f), sum(w) / len(w)
    res[name] = {'dispatches': len(f), 'FETCH_SIZE_mean_KiB': mf, 'WRITE_SIZE_mean_KiB': mw,
                 'hbm_bytes': (2.0 * mf + mw) * 1024.0}
    total += res[name]['hbm_bytes']
res['_derived'] = {'workload': '%s B=%d, %d sweeps + fused marginal read-out' % (workload, batch, sweeps),
                   'hbm_bytes_per_launch': total,
                   'rule': 'separate --pmc passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 summed over the kernels of one sweep call'}
json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
print(json.dumps(res['_derived']))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha          # noqa: E402
key = sys.argv[7] if len(sys.argv) > 7 else workload
reg_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
try:
    reg = json.load(open(reg_path))
except (OSError, ValueError):
    reg = {}
reg[key] = {'hbm_bytes_per_launch': total, 'source': os.path.relpath(out_path, ROOT), 'batch': batch, 'sweeps': sweeps,
            'kernel_sources_sha': kernel_sources_sha()}
json.dump(reg, open(reg_path, 'w'), indent=1, sort_keys=True)
