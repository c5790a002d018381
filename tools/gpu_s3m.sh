#!/bin/bash
# r4 session 3, call M: automatic K2 chunks -- full GPU suite, default bench, two other workloads
out=gpurun_out/s3m; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $out/pytest.log
cat $out/pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_64k_sparse.json 2> $out/bench.err
for w in 128k 32k-8x8 64k-sparse-4x4; do
  timeout 500 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
python - "$out" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"),
              "bwd", (d.get("backward") or {}).get("ms_per_pass"), (d.get("backward") or {}).get("parity", {}).get("rel_rmse"),
              "rt", (d.get("roundtrip") or {}).get("ms_per_pass"))
    except Exception as exc:
        print(f, "FAILED", exc)
PY
