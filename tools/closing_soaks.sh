#!/bin/bash
# at the close of round 4 (multi-envelope pieces, family 4, pointwise thresholds):  bash tools/closing_soaks.sh > gpurun_out/closing_soaks.log
mkdir -p gpurun_out
python tools/fuzz_soak.py 800000 10000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 800000..: /'
python tools/fuzz_soak.py 900000 8000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 900000..: /'
python tools/prims_soak.py 3000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/prims_soak 3000 (grid, float, time list): /'
python tools/chain_soak.py 400000 3000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 400000..: /'
python tools/fmul_soak.py 1200 100000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fmul_soak seeds 100000..: /'
python tools/powers_soak.py 400 9000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/powers_soak seeds 9000..: /'
python tools/erf_soak.py 95000 1500 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 95000..: /'
