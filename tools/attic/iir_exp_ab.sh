for v in ${VARIANTS:-base iirx1 iirx2 iirx3 base}; do for sh in "256 1e7 2" "64 1e7 2" "16 1e7 2"; do
  WFK_IIR_ONEPASS=1 WFK_LIB=$PWD/_ab/libwfk_$v.so python tools/iir_bench.py $sh 2>/dev/null | tail -1 | sed "s/^/$v $sh: /"
done; done
