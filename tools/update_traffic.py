#!/usr/bin/env python3
"""profiles/traffic.json from the committed PMC summaries: what bench.py quotes under
`roofline.traffic` (HBM bytes per launch of the dominant kernel) and, for kernels bound by fp64 VALU
issue rather than by HBM, `roofline_valu` (VALU wave instructions per launch, shader clock).

    python tools/update_traffic.py  workload=profiles/<tag>_<workload>_pmc.json[:kernel]  ...

`kernel` = the short kernel name inside the summary (default: the entry with the largest traffic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, 'profiles', 'traffic.json')
table = json.load(open(out_path)) if os.path.exists(out_path) else {}
for arg in sys.argv[1:]:
    wl, rest = arg.split('=', 1)
    path, _, kern = rest.partition(':')
    d = json.load(open(os.path.join(ROOT, path)))
    ents = {k: v for k, v in d.items() if isinstance(v, dict) and 'traffic_bytes_per_launch' in v}
    k = kern or max(ents, key=lambda k: ents[k]['traffic_bytes_per_launch'])
    c = {n: x['per_launch_mean'] for n, x in ents[k]['counters'].items()}
    ent = {'bytes': ents[k]['traffic_bytes_per_launch'], 'source': os.path.basename(path), 'kernel': k}
    if 'SQ_INSTS_VALU' in c:
        ent['valu_wave_instr'] = c['SQ_INSTS_VALU']
    stats = os.path.join(ROOT, path.replace('_pmc.json', '_kernel_stats.csv'))
    if 'GRBM_GUI_ACTIVE' in c and os.path.exists(stats):
        import csv
        for row in csv.DictReader(open(stats)):
            if k in row['Name']:
                ent['profiled_kernel_ms'] = float(row['AverageNs']) / 1e6
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / kernel time = shader clock
                ent['shader_clock_ghz'] = c['GRBM_GUI_ACTIVE'] / 8 / float(row['AverageNs'])
                break
    # the shape the pass ran on (bench.py's default shape of that workload): bench.py quotes the counters
    # only for a run of the same algorithmic bytes
    sys.path.insert(0, ROOT)
    import bench
    ch, pts = bench.default_shape(wl.split('.')[0])
    elem = 4 if (wl.split('.')[0] == 'c3' or wl.endswith('.f32')) else 8
    ent['algo_bytes'] = ch * pts * elem
    ent['algo_samples'] = ch * pts
    table[wl] = ent
    print(wl, ent)
json.dump(table, open(out_path, 'w'), indent=1)
