#!/bin/bash
# VALU / SALU / LDS instructions and wave cycles per kernel of any python command (one --pmc pass, kernel trace beside it):
#   tools/pmc_valu.sh <tag> <python args...>        e.g. tools/pmc_valu.sh tt tools/awg_shapes_bench.py "ten tones"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmcv_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $out -o run -- python3 "$@" > $out/log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections, re
def short(name):
    m = re.search(r'((?:wfk|iir|fir|spec)_\w+<[^(]*>|(?:wfk|iir|fir|spec)_\w+)', name)
    return m.group(1) if m else None
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
dur = collections.defaultdict(list)
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r['Kernel_Name']): dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, c in acc.items():
    m = len(n[k])
    print(k, 'launches', m, 'ms %.3f' % (sum(dur[k]) / max(1, len(dur[k]))), ' '.join('%s %.4g' % (x, y / m) for x, y in sorted(c.items())))
PY
grep -v amdgpu.ids $out/log | tail -3
