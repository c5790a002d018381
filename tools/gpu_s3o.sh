#!/bin/bash
out=gpurun_out/r4final5; mkdir -p $out
export TMPDIR=/tmp
tools/gpu_pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_kernels.json" "$out/pmc_kernels.json"; cp "$out/pmc/pmc_kernels.txt" "$out/pmc_kernels.txt"; cp "$out/pmc/kernel_stats.txt" "$out/kernel_stats_64k_sparse_serial.txt"; rm -rf "$out/pmc"
cp "$out/pmc_kernels.json" profiles/r4_pmc_kernels.json
cat "$out/pmc_kernels.txt"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_64k_sparse.json" 2> "$out/bench_64k_sparse.err"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4final5/bench_64k_sparse.json"))
print(d["ms_per_step"], d["hbm_algorithmic_frac_of_peak"], d["roofline"]["traffic_build"]["state"])
PY
timeout 300 python -m pytest tests -m gpu -q -x -k "band or bench_shape" 2>&1 | tail -2
