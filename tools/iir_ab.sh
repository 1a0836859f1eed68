#!/bin/bash
# same-box A/B of IIR builds: tools/iir_ab.sh "<variants>"  (libs in _ab/)
for v in $1; do for sh in "64 1e7 2" "64 1e7 1" "256 1e7 2" "256 1e7 1" "16 1e7 2" "1024 1e6 2"; do
  WFK_IIR_ONEPASS=1 WFK_LIB=_ab/libwfk_$v.so python tools/iir_bench.py $sh | sed "s/^/$v /"
done; done
