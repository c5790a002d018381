#!/bin/bash
# same-box A/B of the K1 / finish row kernel: variants/head.so (previous commit) against the working tree's build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k1ab
o=gpurun_out/k1ab
rm -f $o/k1.txt
for i in 1 2; do
  SWIFTLY_HIP_LIB=$PWD/variants/head.so timeout 300 python tools/time_k1_band.py >> $o/k1.txt 2>&1
  timeout 300 python tools/time_k1_band.py >> $o/k1.txt 2>&1
done
grep -v "Warn\|amdgpu.ids" $o/k1.txt
timeout 600 python -m pytest tests/test_hip_backward_parity_gpu.py tests/test_hip_band_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
