#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_band_pipeline_gpu.py tests/test_hip_sizes_gpu.py tests/test_hip_catalogue_sweep_gpu.py -q -x 2>&1 | tail -8 | tee $out/pytest.log
tools/ab_bench.sh $out "new:A=1" "base:SWIFTLY_HIP_LIB=/root/repo/variants/base.so" 2>&1 | tee $out/ab.log
tools/gpu_trace_only.sh $out/trace A=1
