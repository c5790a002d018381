#!/bin/bash
# r4 session 3, call F: backward two-stream overlap (SWIFTLY_PREFETCH=0/1) -- backward tests + bench A/B
out=gpurun_out/s3f; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "backward or roundtrip or delayed or staged or api" 2>&1 | tail -4 > $out/pytest.log
cat $out/pytest.log
for pf in 0 1 0 1; do
  SWIFTLY_PREFETCH=$pf timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_pf$pf.json 2> $out/bench.err
  python - $out/bench_pf$pf.json "$pf" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
b = d.get("backward") or {}
print("prefetch", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "bwd", b.get("ms_per_pass"), "bwd parity", (b.get("parity") or {}).get("rel_rmse"), "roundtrip", (d.get("roundtrip") or {}))
PY
done
