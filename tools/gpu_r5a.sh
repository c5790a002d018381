#!/bin/bash
# r5 session 1: dispatch timeline of the production schedule + baseline numbers on this round's box
out=gpurun_out/r5a; mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$here/$out/kt" -o kt -- python "$here/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-backward > "$here/$out/kt.log" 2>&1 )
db=$(find "$out/kt" -name '*.db' | head -1)
python tools/trace_timeline.py dump "$db" "$out/timeline.csv"
python tools/rocpd_stats.py "$db" > "$out/kernel_stats.txt" 2>&1
rm -rf "$out/kt"
python tools/trace_timeline.py gaps "$out/timeline.csv" | tail -40
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5a/bench.json"))
print("ms_per_step", d["ms_per_step"], "parity", d.get("parity", {}).get("rel_rmse"), {k: v.get("total_ms", v.get("avg_ms")) for k, v in d.get("stages", {}).items()})
PY
timeout 300 python tools/time_k1_band.py 2>&1 | grep -v "Warn\|amdgpu.ids" > $out/k1.txt; cat $out/k1.txt
