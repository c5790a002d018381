#!/bin/bash
# A/B runs of bench.py under tuning-knob environments (one process each: the knobs are read once per process).
# usage: tools/ab_bench.sh OUTDIR "NAME1:ENV1=V1 ENV2=V2" "NAME2:..." ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  rc=$?
  python - "$out/bench_$name.json" "$name" "$rc" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    par = d["parity"]["rel_rmse"] if d.get("parity") else None
    print(sys.argv[2], "rc=" + sys.argv[3], d["ms_per_step"], {k: v.get("total_ms", v.get("avg_ms")) for k, v in d["stages"].items()}, "parity", par)
except Exception as exc:
    print(sys.argv[2], "rc=" + sys.argv[3], "FAILED", exc)
PY
done
