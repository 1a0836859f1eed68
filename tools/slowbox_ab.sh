tools/clock_probe.sh
for v in libm poly libm poly; do
  WFK_LIB=_ab/libwfk_$v.so python bench.py --workload sampler256 --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms'],5), round(d['roofline']['frac'],3))"
done
