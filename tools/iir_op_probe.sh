#!/bin/bash
# phases of the single-pass IIR kernel, same box: 3-launch form, one-pass full, and the probe builds
# (tools/ab_build.sh opK wfk_iir -DOP_PROBE=K: 1 = load/transposes/store only, 2 = + sweeps and scans,
#  no flags, 3 = + publishing, no look-back)
set -e
python tools/iir_bench.py
WFK_IIR_ONEPASS=1 python tools/iir_bench.py
for k in 1 2 3 4 5; do
  [ -f _ab/libwfk_op$k.so ] && WFK_LIB=_ab/libwfk_op$k.so WFK_IIR_ONEPASS=1 python tools/iir_bench.py | sed "s/^/probe $k: /"
done
