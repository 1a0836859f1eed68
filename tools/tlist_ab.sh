#!/bin/bash
# run on the GPU box: the time-list tier over builds of the kernels unit (tools/ab_build.sh <name> wfk_kernels -D...):
#     tools/tlist_ab.sh default name1 name2 ...
# and against itself with the pointwise fusion off (WFK_DISABLE_TLFUSE=1: every factor on device libm, the round-3 tier)
for v in "${@:-default}"; do
  lib=waveforms_amd/csrc/libwfk_hip.so; [ "$v" != default ] && lib=_ab/libwfk_$v.so
  for shape in headline flattop multitone; do
    WFK_LIB=$PWD/$lib python tools/tlist_bench.py $shape 2>/dev/null | sed "s/^/$v /" || exit 1
  done
done
WFK_DISABLE_TLFUSE=1 python tools/tlist_bench.py headline
