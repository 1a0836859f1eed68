#!/bin/bash
# fresh seed ranges beyond tools/all_soaks.sh (round 4, after the last kernel changes):  bash tools/more_soaks.sh > gpurun_out/more_soaks.log
mkdir -p gpurun_out
python tools/fuzz_soak.py 200000 25000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 200000..: /'
python tools/fuzz_soak.py 300000 8000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 300000..: /'
python tools/prims_soak.py 8000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/prims_soak 8000 (grid, float, time list): /'
python tools/mdrag_soak.py 2000 gpu 2>/dev/null | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/mdrag_soak: /'
python tools/erf_soak.py 50000 4000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 50000..: /'
python tools/chain_soak.py 100000 6000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 100000..: /'
python tools/call_api_soak.py 3000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/call_api_soak 3000: /'
