#!/bin/bash
# same-box A/B of chain (sampler fused into the FIR) builds: tools/chain_ab.sh name1 name2 ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=waveforms_amd/csrc/libwfk_hip.so; [ "$v" != default ] && lib=_ab/libwfk_$v.so
  WFK_LIB=$PWD/$lib python bench.py --workload ${WL:-c4} --no-cpu-baseline --no-also --steps 10 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$v', 'kernel_ms %.3f' % l['roofline']['kernel_ms'], l['roofline']['kernel'])"
done
