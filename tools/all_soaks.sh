#!/bin/bash
# the GPU soaks in one call (counts quoted in DESIGN.md 4):  bash tools/all_soaks.sh > gpurun_out/all_soaks.log
mkdir -p gpurun_out   # progress lines go to a file as they come: a call that stays silent for 7 minutes is taken to be hung
python tools/fuzz_soak.py 1000 12000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak (random scripts x grids, f64 \/ f32 \/ tlist vs the C oracle): /'
python tools/fuzz_soak.py 50000 4000 awg | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak awg (random pulse trains on 1-5 GS\/s grids): /'
python tools/erf_soak.py 0 6000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/erf_soak (flat tops around the admission limit): /'
python tools/fuzz_soak_big.py | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/fuzz_soak_big (long grids, offsets): /'
python tools/stage_soak.py 1500 2>/dev/null | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/stage_soak (FIR \/ IIR \/ chain vs SciPy): /'
python tools/call_api_soak.py | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/call_api_soak (drop-in calls): /'
python tools/sample_api_soak.py 2>/dev/null | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/sample_api_soak: /'
python tools/spectral_soak.py 2>/dev/null | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/spectral_soak: /'
python tools/chain_soak.py 0 10000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/chain_soak (random pulse trains through the sampler -> FIR chain): /'
