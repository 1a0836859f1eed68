#!/bin/bash
# run on the GPU box: tools/samp_ab.sh "<variants>" "<workload args;...>"
IFS=';' read -ra WLS <<< "$2"
for v in $1; do for w in "${WLS[@]}"; do
  WFK_LIB=_ab/libwfk_$v.so python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:12], d['dtype'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done
