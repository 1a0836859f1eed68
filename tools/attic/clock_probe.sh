#!/bin/bash
# shader / fabric / memory clocks WHILE the headline kernel runs (rocm-smi sampled next to a 400-step bench)
python bench.py --workload sampler256 --steps 4000 --no-cpu-baseline --no-also > gpurun_out/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 11
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Average Graphics Package Power\|Socket\|junction\|memory)" | tr -s ' \t' ' ' | tr '\n' ';' | cut -c1-400
  echo
  sleep 1
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/clock_probe_bench.json')); print('kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
