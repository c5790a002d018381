#!/bin/bash
# r4 session 3, call H: K1 with the window by LDS-DMA (SWIFTLY_ROW_WLDS=0/1) -- K1 timing, parity tests, bench
out=gpurun_out/s3h; mkdir -p $out; rm -f $out/*.txt
export TMPDIR=/tmp
for v in 0 1 0 1; do
  echo "== SWIFTLY_ROW_WLDS=$v" >> $out/k1.txt
  SWIFTLY_ROW_WLDS=$v timeout 200 python tools/time_k1_band.py 2>&1 | grep "K1" >> $out/k1.txt
done
cat $out/k1.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "band or bench_shape or forward or golden or long_rows or cooperative or virtual" 2>&1 | tail -4 > $out/pytest.log
cat $out/pytest.log
for v in 0 1; do
  SWIFTLY_ROW_WLDS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-backward > $out/bench_$v.json 2> $out/bench.err
  python - $out/bench_$v.json "$v" <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("wlds", sys.argv[2], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity", (d.get("parity") or {}).get("rel_rmse"), "K1", d["stages"]["K1_back_to_back"])
PY
done
