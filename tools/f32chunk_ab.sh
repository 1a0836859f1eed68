for cap in 8 10 12 16 20 24 32 12 16; do
  for w in c3 "sampler256 --dtype f32"; do
  WFK_TPC_F32=$cap python bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cap$cap', d['config']['workload'][:12], d['dtype'], round(d['roofline']['kernel_ms'],5), round(d['roofline']['frac'],3))"
  done
done
