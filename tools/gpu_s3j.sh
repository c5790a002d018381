#!/bin/bash
# r4 session 3, call J: env-knob sweep on the default bench (four-step split, facets per launch group, chunks under the y2-major layout)
out=gpurun_out/s3j; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward > $out/bench.json 2> $out/bench.err
  python - $out/bench.json "$*" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "| ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"), "K2", d["stages"]["K2_wave_facet_transform"]["total_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run X=0
run SWIFTLY_L1_BIAS=1
run SWIFTLY_L1_BIAS=-1
run SWIFTLY_K2_FACETS=3
run SWIFTLY_K2_CHUNK=512,3
run X=0
run SWIFTLY_K2_CHUNK=256,2
run SWIFTLY_K2_CHUNK=512
