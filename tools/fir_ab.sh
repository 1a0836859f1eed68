#!/bin/bash
# run on the GPU box: tools/fir_ab.sh "<f64 variants>" "<f32 variants>" [test-variant]
for v in $1; do WFK_LIB=_ab/libwfk_$v.so python tools/fir_bench.py 256 1e7 1024 f64 2>/dev/null; done
for v in $2; do WFK_LIB=_ab/libwfk_$v.so python tools/fir_bench.py 256 1e7 1024 f32 2>/dev/null; done
if [ -n "$3" ]; then WFK_LIB=_ab/libwfk_$3.so python -m pytest tests/test_gpu_fir.py -q -x 2>&1 | tail -3; fi
