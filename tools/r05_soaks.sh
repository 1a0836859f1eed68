#!/bin/bash
# round 5 (sampler inside the IIR scan, short-tier op families, packed-fp32 float launches, block-parallel compile,
# native flatten):  bash tools/r05_soaks.sh > gpurun_out/r05_soaks.log
mkdir -p gpurun_out
python tools/iirchain_soak.py 0 1200 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/iirchain_soak seeds 0..: /'
python tools/fuzz_soak.py 1100000 6000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg seeds 1100000..: /'
python tools/fuzz_soak.py 1200000 4000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak seeds 1200000..: /'
python tools/chain_soak.py 500000 1500 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak seeds 500000..: /'
python tools/stage_soak.py | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/stage_soak: /'
python tools/erf_soak.py 105000 800 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak seeds 105000..: /'
python tools/fmul_soak.py 600 200000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fmul_soak seeds 200000..: /'
