#!/bin/bash
# r4 session 3, call P: K3 ahead with K2 in the planned-wave prefetch (SWIFTLY_PREFETCH=2) against K2 only (=1)
out=gpurun_out/s3p; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-backward $EXTRA > $out/bench.json 2> $out/bench.err
  python - $out/bench.json "$*" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "| ms/step", d["ms_per_step"], "parity", (d.get("parity") or {}).get("rel_rmse"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
run SWIFTLY_PREFETCH=2
EXTRA=--no-verify
for rep in 1 2 3; do
  run SWIFTLY_PREFETCH=1
  run SWIFTLY_PREFETCH=2
done
