#!/bin/bash
# r4 session 3, call N: chunked two-stream four-step for the backward mirror (gather-sum load + column scatter)
out=gpurun_out/s3n; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "backward or roundtrip or accumulate or adjoint" 2>&1 | tail -4 > $out/pytest.log
cat $out/pytest.log
for rep in 1 2; do
for ch in 0 auto; do
  if [ "$ch" = auto ]; then unset SWIFTLY_K2_CHUNK; else export SWIFTLY_K2_CHUNK=$ch; fi
  AXES=1 REPS=4 timeout 300 python tools/roundtrip_64k.py > $out/rt.log 2>&1
  echo "chunk $ch: $(grep 'backward ms' $out/rt.log | awk '{print $NF}' | tr '\n' ' ')" | tee -a $out/ab.txt
done
done
