#!/bin/bash
# Which of the headline kernel's rates (3.16 / 3.44 / 3.58 ms on the same binary) does a process get, and
# does it follow the GPU's clock / power state?  12 fresh processes back to back, rocm-smi in between.
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|Power\|Temperature (Sensor junction)\|memory)" | tr -s ' ' | tr '\n' ';' | cut -c1-420
  echo
  python bench.py --workload sampler256 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i kernel_ms', round(d['roofline']['kernel_ms'],4), 'step_ms', round(d['ms_per_step'],4))"
done
