#!/bin/bash
# same-box A/B of lean-kernel builds over the workloads its per-op cost shows in: tools/state_ab.sh "<variants>"  (libs in _ab/)
for v in $1; do
  export WFK_LIB=_ab/libwfk_$v.so
  for w in sampler256 c3 "sampler256 --dtype f32" c2 far; do
    python bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:12], d['dtype'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
  done
  for nt in 4 10; do python tools/multitone_bench.py $nt | sed "s/^/$v /"; done
  python tools/readout_bench.py | sed "s/^/$v /"
done
