#!/bin/bash
# after the round's last short-tier changes (sampled flat-top edges, bare-carrier loop, blocks with tables):
#   bash tools/r05_final_soaks.sh > gpurun_out/r05_final_soaks.log
mkdir -p gpurun_out
python tools/flattop_awg_soak.py 1500 700000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/flattop_awg_soak seeds 700000..: /'
python tools/fuzz_soak.py 1300000 6000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 1300000..: /'
python tools/fuzz_soak.py 1400000 3000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 1400000..: /'
python tools/erf_soak.py 115000 800 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 115000..: /'
python tools/fmul_soak.py 600 210000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fmul_soak seeds 210000..: /'
python tools/chain_soak.py 510000 1000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 510000..: /'
python tools/iirchain_soak.py 5000 400 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/iirchain_soak seeds 5000..: /'
