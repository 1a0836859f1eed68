#!/bin/bash
# run on the GPU box: the time-list tier, this tree against the round-3 kernels (_ab/libwfk_base.so: tools/ab_build.sh
# style, wfk_kernels.hip of the previous commit) and against itself with the pointwise fusion off
for shape in headline flattop multitone readme; do
  WFK_LIB=_ab/libwfk_base.so WFK_DISABLE_TLFUSE=1 python tools/tlist_bench.py $shape || exit 1
  WFK_DISABLE_TLFUSE=1 python tools/tlist_bench.py $shape || exit 1
  python tools/tlist_bench.py $shape || exit 1
done
