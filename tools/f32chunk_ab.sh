for pass in 1 2 3; do for e in "X=1" "WFK_TPC_F32=8" "WFK_TPC_F32=20"; do for w in c3 "sampler256 --dtype f32"; do
 env $e python bench.py --workload $w --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['config']['workload'].split(':')[0], d['dtype'], round(d['roofline']['kernel_ms'],5))"
done; done; done
