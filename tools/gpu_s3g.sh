#!/bin/bash
out=gpurun_out/s3g; mkdir -p $out; rm -f $out/k1.txt
for v in "" 1 "" 1; do
  echo "== SWIFTLY_ROW_NOWIN=$v" >> $out/k1.txt
  if [ -n "$v" ]; then export SWIFTLY_ROW_NOWIN=1; else unset SWIFTLY_ROW_NOWIN; fi
  timeout 200 python tools/time_k1_band.py 2>&1 | grep "K1" >> $out/k1.txt
done
cat $out/k1.txt
