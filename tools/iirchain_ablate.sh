#!/bin/bash
# VALU instructions per sample and kernel time of A/B builds of the sampler -> IIR chain (libs in _ab/, tools/ab_build.sh):
#   tools/iirchain_ablate.sh "<variants>" [workload]
# one --pmc pass per variant (kernel trace only beside it), on the GPU box
export TMPDIR=/tmp
WL=${2:-iir_chain}
for v in $1; do
  out=gpurun_out/ablate_$v
  rm -rf $out; mkdir -p $out
  WFK_LIB=_ab/libwfk_$v.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out -o run -- python3 bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-also > $out/log 2>&1
  python3 - $v $out <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'iir_' in r['Kernel_Name'] or 'wfk_' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
dur = collections.defaultdict(list)
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, c in acc.items():
    m = len(n[k])
    print(v, k, 'launches', m, 'ms %.3f' % (sum(dur[k]) / max(1, len(dur[k]))), ' '.join('%s %.4g' % (x, y / m) for x, y in sorted(c.items())),
          'VALU/sample %.1f' % (c['SQ_INSTS_VALU'] / m * 64 / 2.56e9))
PY
done
