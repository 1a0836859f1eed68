#!/bin/bash
# fresh seed ranges after the round's last kernel changes (closing multipliers, tone loop):  bash tools/final_soaks.sh > gpurun_out/final_soaks.log
mkdir -p gpurun_out
python tools/fuzz_soak.py 400000 20000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 400000..: /'
python tools/fuzz_soak.py 500000 8000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 500000..: /'
python tools/prims_soak.py 6000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/prims_soak 6000 (grid, float, time list): /'
python tools/erf_soak.py 70000 3000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 70000..: /'
python tools/chain_soak.py 200000 4000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 200000..: /'
python tools/call_api_soak.py 2000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/call_api_soak 2000: /'
python tools/fmul_soak.py 1500 40000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fmul_soak seeds 40000..: /'
python tools/fuzz_soak_big.py | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak_big (long grids, offsets): /'
