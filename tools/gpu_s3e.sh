#!/bin/bash
# r4 session 3, call E: y2-major four-step intermediate (SWIFTLY_SCRATCH_LAYOUT) -- tests + bench A/B
out=gpurun_out/s3e; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $out/pytest.log
cat $out/pytest.log
for lay in 0 1 0 1; do
  SWIFTLY_SCRATCH_LAYOUT=$lay timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_lay$lay.json 2> $out/bench.err
  python - $out/bench_lay$lay.json "$lay" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("layout", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"), "K2 stage", d["stages"]["K2_wave_facet_transform"]["total_ms"], "K345", d["stages"]["K345_extract_sum_finish"]["total_ms"], "bwd", (d.get("backward") or {}).get("ms_per_pass"))
PY
done
