#!/bin/bash
# after the last short-tier / pointwise changes of round 4:  bash tools/last_soaks.sh > gpurun_out/last_soaks.log
mkdir -p gpurun_out
python tools/fuzz_soak.py 600000 12000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 600000..: /'
python tools/fuzz_soak.py 700000 8000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 700000..: /'
python tools/prims_soak.py 4000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/prims_soak 4000 (grid, float, time list): /'
python tools/chain_soak.py 300000 3000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 300000..: /'
python tools/fmul_soak.py 900 60000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fmul_soak seeds 60000..: /'
python tools/powers_soak.py 600 5000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/powers_soak seeds 5000..: /'
python tools/erf_soak.py 90000 2000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 90000..: /'
