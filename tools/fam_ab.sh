#!/bin/bash
# same-box A/B around the split of the lean kernel into op families: headline, fp32, C3 (bench.py also),
# 10 tones, flat-top readout, chirps -- default library against _ab/libwfk_head.so (the commit before)
for lib in "" _ab/libwfk_head.so ""; do
  echo "== ${lib:-default}"
  WFK_LIB=$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])
for k in ('f32','c2','c3','far'):
    print(k, round(d['also'][k]['kernel_ms'],5), d['also'][k]['kernel'])
print('c4', round(d['also']['c4']['step_ms'],3))"
  WFK_LIB=$lib python tools/multitone_bench.py 10 2>/dev/null | tail -2
  WFK_LIB=$lib python tools/readout_bench.py 2>/dev/null | tail -2
  WFK_LIB=$lib python tools/direct_tier_bench.py 2>/dev/null | grep -E "chirp|sinc"
done
