#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline figures are judged against.  Runs ON the
# GPU box (through gpurun) from the repo root:
#
#     bash tools/collect_profiles.sh <tag> [workload]      # e.g. r01 c4
#
# 1. kernel-trace + stats pass of `bench.py --workload <workload>`  -> per-kernel durations
# 2. separate --pmc passes (WRITE_SIZE | FETCH_SIZE | SQ set), each with --kernel-trace only,
#    as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
# 3. tools/summarize_profiles.py folds the CSVs into gpurun_out/profiles/<tag>_<workload>_*.{csv,json}
# The program itself follows `--` (python3, no env/bash hop).
set -e -o pipefail
TAG=${1:?tag}
WL=${2:-c4}
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT" gpurun_out/profiles
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-also"   # (--no-also: the c2 / c3 / c4 legs of the default line run the same kernel symbols; each is profiled by its own workload)
# stage benches outside bench.py (their own scripts): iir, readout, multitone
case "$WL" in
  iir) ARGS="tools/iir_bench.py" ;;
  readout) ARGS="tools/readout_bench.py" ;;
  multitone) ARGS="tools/multitone_bench.py 10" ;;
  awg_f32) ARGS="bench.py --workload awg --dtype f32 --steps 5 --warmup 1 --no-cpu-baseline --no-also" ;;
esac
echo "== stats pass ($WL)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ARGS > "$OUT/stats.log" 2>&1
for grp in WRITE_SIZE FETCH_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
    name=$(echo $grp | cut -d' ' -f1)
    echo "== pmc pass $name"
    # (a counter this part does not have fails its own pass only: the stall breakdown of round 4 is best effort)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o run -- python3 $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "   pass $name failed (see $OUT/pmc_$name.log)"
done
python3 tools/summarize_profiles.py "$OUT" "gpurun_out/profiles/${TAG}_${WL}"
grep "^{" "$OUT/stats.log" | tail -1 > "gpurun_out/profiles/${TAG}_${WL}_bench.json" || rm -f "gpurun_out/profiles/${TAG}_${WL}_bench.json"
