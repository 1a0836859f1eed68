#!/bin/bash
# same-box sweep of tiles per chunk (WFK_TPC) of the double lean launches
for pass in 1 2 3; do for tpc in 8 5; do for w in sampler256 c5 far; do
  WFK_TPC=$tpc python bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tpc$tpc', d['config']['workload'].split(':')[0], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done; done; done
