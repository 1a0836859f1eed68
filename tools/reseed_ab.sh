#!/bin/bash
# same-box A/B: exact-reseed interval of the lean kernel x tiles per chunk (libs _ab/libwfk_reseedN.so)
for v in cur:8 cur:6 reseed16:8 reseed16:12 reseed16:16 reseed32:16 reseed32:24 reseed32:32 cur:8; do
  lib=${v%%:*}; tpc=${v##*:}
  for w in sampler256 "sampler256 --dtype f32" c3; do
  WFK_TPC=$tpc WFK_LIB=_ab/libwfk_$lib.so python bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib tpc$tpc', d['config']['workload'][:12], d['dtype'], round(d['roofline']['kernel_ms'],5), round(d['roofline']['frac'],3), 'err', d.get('max_abs_err_vs_numpy_ref'))"
  done
done
