#!/bin/bash
# r5 session 3: new bench.py keys (backward median + kernels, other_workloads), counter summaries of this build for both directions
out=gpurun_out/r5c; mkdir -p $out
export TMPDIR=/tmp
tools/gpu_pmc.sh "$out/pmc" 64k-sparse forward > "$out/pmc_fwd.log" 2>&1
tools/gpu_pmc.sh "$out/pmc" 64k-sparse backward > "$out/pmc_bwd.log" 2>&1
cp "$out/pmc/pmc_kernels.json" profiles/r5_pmc_kernels.json
cat "$out/pmc/pmc_kernels_forward.txt" "$out/pmc/pmc_kernels_backward.txt"
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c/bench.json"))
print("ms_per_step", d["ms_per_step"], d["hbm_algorithmic_frac_of_peak"], d["roofline"]["traffic_build"]["state"])
b = d["backward"]; print("backward", b["ms_per_pass"], b["each_ms"], b["warmup_ms"], (b.get("roofline") or {}).get("frac"))
for r in (b.get("kernels") or {}).get("rows", []): print("  ", r["stage"], r["us_per_pass"], r["frac_algorithmic"], r["frac_sustained"])
print("other", json.dumps(d.get("other_workloads"), indent=None)[:1200])
print("hp", d["high_precision"]["ms_per_step"], d["high_precision"]["parity"]["rel_rmse"])
PY
