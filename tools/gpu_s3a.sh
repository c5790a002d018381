#!/bin/bash
# r4 session 3, call A: planned-wave prefetch A/B (SWIFTLY_PREFETCH=0/1) + forward API tests
out=gpurun_out/s3a; mkdir -p $out
export TMPDIR=/tmp
for pf in 0 1 0 1; do
  SWIFTLY_PREFETCH=$pf timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward > $out/bench_pf$pf.json 2> $out/bench_pf$pf.err
  python - $out/bench_pf$pf.json $pf <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("prefetch", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"))
PY
done
timeout 600 python -m pytest tests -m gpu -q -x -k "forward or api or band or plan" 2>&1 | tail -5 > $out/pytest.log
cat $out/pytest.log
