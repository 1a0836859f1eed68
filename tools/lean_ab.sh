#!/bin/bash
# same-box A/B of a libwfk build on the default bench line (headline + also):  tools/lean_ab.sh <lib>
for lib in "" "$@" ""; do
  echo "== ${lib:-default}"
  WFK_LIB=$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))
for k in ('f32','c2','c3','far','awg','awg_duty30'):
    print(k, round(d['also'][k]['kernel_ms'],5), round(d['also'][k]['frac'],4))
print('c4', round(d['also']['c4']['step_ms'],3))"
done
