#!/bin/bash
# r4 session 3, call K: interleaved repeats (box noise is +-0.5 ms between processes)
out=gpurun_out/s3k; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-backward --no-verify > $out/bench.json 2> $out/bench.err
  python - $out/bench.json "$*" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "| ms/step", d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run X=0
for rep in 1 2 3 4; do
  run X=0
  run SWIFTLY_K2_CHUNK=256,2
  run SWIFTLY_K2_CHUNK=512
  run SWIFTLY_L1_BIAS=1
  run SWIFTLY_K2_CHUNK=256,3
done
