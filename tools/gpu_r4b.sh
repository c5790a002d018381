#!/bin/bash
out=gpurun_out/r4b; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/k1_trace.py 2>&1 | tail -40 | tee $out/trace.log
for v in prio2 prio0; do
  SWIFTLY_HIP_LIB=/root/repo/variants/$v.so timeout 200 python tools/time_k1_band.py 2>&1 | tail -3
done | tee $out/k1.log
timeout 200 python tools/time_k1_band.py 2>&1 | tail -3 | tee -a $out/k1.log
