#!/bin/bash
# r4 session 2, call A: column-tile copy ceilings, Infinity-Cache experiments on the per-wave stages, bench at HEAD
out=gpurun_out/s2a; mkdir -p $out
export TMPDIR=/tmp
timeout 120 tools/colcopy_bw.bin > $out/colcopy.txt 2>&1
timeout 200 python tools/exp_mall.py > $out/mall_default.txt 2>&1
SWIFTLY_HIP_LIB=$PWD/variants/ntoff.so timeout 200 python tools/exp_mall.py > $out/mall_ntoff.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/colcopy.txt; tail -12 $out/mall_default.txt; tail -12 $out/mall_ntoff.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/s2a/bench.json"))
print("ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"), "bwd", (d.get("backward") or {}).get("ms_per_pass"))
print({k: v.get("total_ms", v.get("avg_ms")) for k, v in d["stages"].items()})
PY
