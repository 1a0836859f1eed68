#!/bin/bash
# every bench line DESIGN.md quotes, in one GPU call:  bash tools/all_benches.sh > gpurun_out/all_benches.log
python bench.py > gpurun_out/bench_default.json && cut -c1-400 gpurun_out/bench_default.json
for w in "c4" "c3" "c2" "c2_duty30" "c2_drag" "sampler256 --dtype f32"; do
  python bench.py --workload $w --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['dtype'], 'value %.0f Msamples/s' % d['value'], 'ms/step %.4f' % d['ms_per_step'], r['kernel'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'sampler_ms', r.get('sampler_kernel_ms'))"
done
python tools/iir_bench.py 2>/dev/null
python tools/iir_bench.py 256 1e7 2 2>/dev/null
python tools/iir_bench.py 256 1e7 1 2>/dev/null
WFK_IIR_ONEPASS=0 python tools/iir_bench.py 256 1e7 2 2>/dev/null | sed 's/^/three-launch form: /'
python tools/fir_bench.py 256 1e7 1024 f64 2>/dev/null
python tools/fir_bench.py 256 1e7 1024 f32 2>/dev/null
python tools/stream_ceiling.py 2>/dev/null
python tools/multitone_bench.py 10 2>/dev/null
python tools/readout_bench.py 2>/dev/null
python tools/direct_tier_bench.py 2>/dev/null
python tools/tlist_bench.py 2>/dev/null
