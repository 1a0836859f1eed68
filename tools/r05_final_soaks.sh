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
# the drop-in API layers (sample(filters=) now runs through wfk_chain_iir_*), constructors, powers, spectral ops
python tools/sample_api_soak.py 600 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/sample_api_soak: /'
python tools/call_api_soak.py 1500 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/call_api_soak: /'
python tools/prims_soak.py 3000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/prims_soak gpu: /'
python tools/mdrag_soak.py 1000 gpu | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/mdrag_soak gpu: /'
python tools/powers_soak.py 900 30000 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/powers_soak seeds 30000..: /'
python tools/spectral_soak.py 200 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/spectral_soak: /'
python tools/stage_soak.py 300 | tee -a gpurun_out/soak_progress.log | tail -n 1 | sed 's/^/stage_soak: /'
