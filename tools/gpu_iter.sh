#!/bin/bash
# One measurement iteration on the GPU box:  tools/gpu_iter.sh OUTDIR [pytest -k expression | "all" | "none"]
#   bench.py (10 steps) -> OUTDIR/bench.json ; rocprofv3 --kernel-trace of a 2-step bench -> OUTDIR/kernel_stats.txt
out=$1; sel=${2:-none}
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$sel" = "all" ]; then
  timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$out/pytest.log"
elif [ "$sel" != "none" ]; then
  timeout 900 python -m pytest tests -m gpu -q -k "$sel" 2>&1 | tail -40 > "$out/pytest.log"
fi
timeout 400 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS:-} > "$out/bench.json" 2> "$out/bench.err"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$OLDPWD/$out/kt" -o kt -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-backward ${BENCH_ARGS:-} > "$OLDPWD/$out/kt.log" 2>&1 )
db=$(find "$out/kt" -name '*.db' | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats.txt" 2>&1
rm -rf "$out/kt"
[ -f "$out/pytest.log" ] && tail -3 "$out/pytest.log"
python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"),
      "bwd", (d.get("backward") or {}).get("ms_per_pass"), (d.get("backward") or {}).get("parity", {}).get("rel_rmse"))
print({k: v.get("total_ms", v.get("avg_ms")) for k, v in d["stages"].items()})
PY
cut -c1-200 "$out/kernel_stats.txt" | head -16
