#!/bin/bash
# same-box A/B of sampler builds on the float workloads: tools/f32_ab.sh name1 name2 ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=waveforms_amd/csrc/libwfk_hip.so; [ "$v" != default ] && lib=_ab/libwfk_$v.so
  c3=$(WFK_LIB=$PWD/$lib python bench.py --workload c3 --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('%.4f ms %.3f' % (l['roofline']['kernel_ms'], l['roofline']['frac']))")
  f32=$(WFK_LIB=$PWD/$lib python bench.py --dtype f32 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('%.4f ms %.3f' % (l['roofline']['kernel_ms'], l['roofline']['frac']))")
  echo "$v  c3 $c3   f32 $f32"
done
