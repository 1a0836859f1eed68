#!/bin/bash
# units-per-chunk sweep of the short tier on the AWG workload (same box)
for upc in 2 4 8 16 32; do
  echo "== WFK_SH_UPC=$upc"
  WFK_SH_UPC=$upc python tools/awg_bench.py 2048 1e5 2 2>/dev/null | grep -E "float64|float32"
done
