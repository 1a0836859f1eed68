#!/usr/bin/env python3
"""profiles/INDEX.md: per workload, the ONE stats / PMC pair that is current (what profiles/traffic.json -- and through it
bench.py's roofline.traffic / roofline_valu -- trusts), with the figures read from those files.
    python tools/profiles_index.py"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')
t = json.load(open(os.path.join(P, 'traffic.json')))
rows = []
for wl, v in t.items():
    stem = v['source'].replace('_pmc.json', '')
    stats = stem + '_kernel_stats.csv'
    ms = calls = None
    if os.path.exists(os.path.join(P, stats)):
        for r in csv.DictReader(open(os.path.join(P, stats))):
            if v['kernel'] in r['Name']:
                ms, calls = float(r['AverageNs']) / 1e6, int(r['Calls'])
                break
    valu = v.get('valu_wave_instr')
    rows.append((wl, v['kernel'], stats if ms is not None else '-', v['source'],
                 '-' if ms is None else '%.4g ms x %d' % (ms, calls),
                 '%.4g GB = %.3fx' % (v['bytes'] / 1e9, v['bytes'] / v['algo_bytes']),
                 '-' if valu is None else '%.1f' % (valu * 64 / v['algo_samples'])))
with open(os.path.join(P, 'INDEX.md'), 'w') as f:
    f.write('# profiles/ -- which files are current\n\n'
            'One rocprofv3 `--kernel-trace --stats` summary and one summary of the separate `--pmc` passes per workload\n'
            '(`tools/collect_profiles.sh <tag> <workload>`; HBM bytes = WRITE_SIZE + 2 x FETCH_SIZE in KiB, per launch, as\n'
            'MI355X_MICROARCH.md prescribes).  `traffic.json` (written by `tools/update_traffic.py`) is what `bench.py` reads;\n'
            'this table is generated from it (`tools/profiles_index.py`).  Kernels that did not change keep the profile of the\n'
            'round that last touched them.  Everything superseded is under `archive/` (rounds 1-4, five generations of some\n'
            'workloads); `workloads.md` says what each bench key is.\n\n'
            '| workload (bench key) | dominant kernel | stats | PMC summary | kernel time (profiled) | HBM traffic per launch vs algorithmic | VALU instr / sample |\n'
            '|---|---|---|---|---|---|---|\n')
    for r in rows:
        f.write('| ' + ' | '.join('`%s`' % x if i in (2, 3) and x != '-' else x for i, x in enumerate(r)) + ' |\n')
    f.write('\nOther files: `r05_*_bench.json` = the bench line of the profiled run (kernel times under tracing are 3-8 % above\n'
            'un-profiled ones); `r04_closing_soaks.log` = the last soak log of round 4 (the soak tools under `tools/`).\n')
print(open(os.path.join(P, 'INDEX.md')).read())
