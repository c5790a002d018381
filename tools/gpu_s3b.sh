#!/bin/bash
# r4 session 3, call B: K1 row prefetch distance sweep (SWIFTLY_ROW_PF)
out=gpurun_out/s3b; mkdir -p $out; rm -f $out/k1.txt
for pf in 0 -1 0 -1; do
  echo "== SWIFTLY_ROW_PF=$pf" >> $out/k1.txt
  SWIFTLY_ROW_PF=$pf timeout 200 python tools/time_k1_band.py 2>&1 | grep "K1\|finish" >> $out/k1.txt
done
cat $out/k1.txt
