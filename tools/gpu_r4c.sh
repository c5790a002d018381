#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_band_pipeline_gpu.py tests/test_hip_bench_shape_gpu.py tests/test_hip_backward_parity_gpu.py -q -x 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log
tools/ab_bench.sh $out "f64:SWIFTLY_COL_F64=1" "f32:SWIFTLY_COL_F64=0" 2>&1 | tee $out/ab.log
python - <<'PY'
import json
for n in ("f64","f32"):
    d=json.load(open(f"gpurun_out/r4c/bench_{n}.json"))
    print(n, "parity", d["parity"], "\n   backward", {k:v for k,v in d.get("backward",{}).items() if k in ("ms_per_pass","parity")})
PY
