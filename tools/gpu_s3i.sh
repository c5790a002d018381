#!/bin/bash
# r4 session 3, call I: side-stream priority of the planned-wave prefetch
out=gpurun_out/s3i; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
for pr in 0 -1 0 -1; do
  SWIFTLY_SIDE_PRIO=$pr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward --no-verify > $out/bench.json 2> $out/bench.err
  python - $out/bench.json "$pr" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("side prio", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"])
PY
done
