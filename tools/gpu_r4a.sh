#!/bin/bash
# r4 iteration A: parity of the segment-skipping row kernels, K1 A/B over build variants, K2 reverse-order knob
out=gpurun_out/r4a; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_band_pipeline_gpu.py tests/test_hip_backward_parity_gpu.py -q -x -k "prepare_facet_band or finish_facet_band or rows_match_oracle" 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log
for v in base prio0 ntst; do
  SWIFTLY_HIP_LIB=/root/repo/variants/$v.so timeout 200 python tools/time_k1_band.py 2>&1 | tail -3
done | tee $out/k1.log
timeout 200 python tools/time_k1_band.py 2>&1 | tail -3 | tee -a $out/k1.log
SWIFTLY_ROW_SEGSKIP=0 timeout 200 python tools/time_k1_band.py 2>&1 | tail -3 | tee -a $out/k1.log
tools/ab_bench.sh $out "new:A=1" "base:SWIFTLY_HIP_LIB=/root/repo/variants/base.so" "rev:SWIFTLY_K2_REV=1" "revc:SWIFTLY_K2_REV=1 SWIFTLY_SCRATCH_NT=0" 2>&1 | tee $out/ab.log
