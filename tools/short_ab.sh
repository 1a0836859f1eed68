#!/bin/bash
# same-box A/B of libwfk builds on the AWG workload:  tools/short_ab.sh <lib>...   (default lib first, twice)
for lib in "" "$@" ""; do
  echo "== ${lib:-default}"
  WFK_LIB=$lib python tools/awg_bench.py 2048 1e5 2 2>/dev/null | grep -E "float64|float32"
  WFK_LIB=$lib python tools/awg_bench.py 2048 1e5 2 1 2>/dev/null | grep -E "float64"
done
