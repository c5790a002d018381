#!/bin/bash
# r4 session 3, call Q: inter-phase twiddle values requested before the exchange (variants/nowtasm.so, -DSWF_TW_PRELOAD=1)
out=gpurun_out/s3r; mkdir -p $out; rm -f $out/k1.txt
for rep in 1 2; do
  for v in nowtasm default; do
    echo "== $v" >> $out/k1.txt
    if [ $v = default ]; then unset SWIFTLY_HIP_LIB; else export SWIFTLY_HIP_LIB=$PWD/variants/nowtasm.so; fi
    timeout 200 python tools/time_k1_band.py 2>&1 | grep "K1\|finish" >> $out/k1.txt
  done
done
cat $out/k1.txt
