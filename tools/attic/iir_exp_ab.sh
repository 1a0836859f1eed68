for v in ${VARIANTS:-base}; do for sh in "256 1e7 2" "256 1e7 1" "64 1e7 2" "256 1e7 4 first"; do
  WFK_IIR_ONEPASS=1 WFK_LIB=$PWD/_ab/libwfk_$v.so python tools/iir_bench.py $sh 2>/dev/null | tail -1 | sed "s/^/$v $sh: /"
done; done
