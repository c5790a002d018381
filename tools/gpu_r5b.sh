#!/bin/bash
# r5 session 2: K1 with one asm statement per complex product; tile-major four-step scratch
out=gpurun_out/r5b; mkdir -p $out
for i in 1 2; do
  SWIFTLY_HIP_LIB=$PWD/variants/oneasm_row.so timeout 300 python tools/time_k1_band.py 2>&1 | grep "ms per facet" >> $out/k1.txt
  timeout 300 python tools/time_k1_band.py 2>&1 | grep "ms per facet" >> $out/k1.txt
done
cat $out/k1.txt
export BENCH_ARGS="--no-backward"
tools/ab_bench.sh $out "base1:SWIFTLY_X=0" "tile1:SWIFTLY_SCRATCH_TILE=1" "base2:SWIFTLY_X=0" "tile2:SWIFTLY_SCRATCH_TILE=1"
