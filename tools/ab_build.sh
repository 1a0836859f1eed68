#!/bin/bash
# A/B builds of one translation unit of libwfk_hip.so:  tools/ab_build.sh <name> <unit> <flags...>
#   e.g. tools/ab_build.sh notw wfk_fir_fused -DWFK_FIR_EXP=1
# -> _ab/libwfk_<name>.so (git-ignored; travels with gpurun).  Select with WFK_LIB=_ab/libwfk_<name>.so
set -e
cd "$(dirname "$0")/../waveforms_amd/csrc"
name=$1; unit=$2; shift 2
mkdir -p ../../_ab
make -s
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -I../../include -I. --offload-arch=gfx950 -c $unit.hip -o ../../_ab/${unit}_$name.o "$@"
objs=$(ls _obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_ab/libwfk_$name.so $objs ../../_ab/${unit}_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
echo built _ab/libwfk_$name.so
