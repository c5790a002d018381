#!/bin/bash
# r4 session 3, call C: chunked two-stream four-step of K2 (SWIFTLY_K2_CHUNK) -- parity tests + bench A/B
out=gpurun_out/s3c; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
SWIFTLY_K2_CHUNK=256 timeout 600 python -m pytest tests -m gpu -q -x -k "band or bench_shape or forward" 2>&1 | tail -4 > $out/pytest_chunk256.log
cat $out/pytest_chunk256.log
for ch in 0 256 512 128 0 256 "512,3" "256,2"; do
  SWIFTLY_K2_CHUNK=$ch timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward > $out/bench.json 2> $out/bench.err
  python - $out/bench.json "$ch" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("chunk", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"), "K2 stage", d["stages"]["K2_wave_facet_transform"]["total_ms"])
PY
done
