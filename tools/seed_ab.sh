#!/bin/bash
# same-box A/B of the lean kernel's seed arithmetic: libm (-DWFK_SEED_LIBM) vs the inline polynomials
for v in libm poly libm poly libm poly; do
  for w in sampler256 "sampler256 --dtype f32" c3 c2 far; do
  WFK_LIB=_ab/libwfk_$v.so python bench.py --workload $w --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:12], d['dtype'], round(d['roofline']['kernel_ms'],5), round(d['roofline']['frac'],3), 'err', d.get('max_abs_err_vs_numpy_ref'))"
  done
done
