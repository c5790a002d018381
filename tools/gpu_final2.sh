#!/bin/bash
# The round's final collection, trimmed (r4 session 3):  tools/gpu_final2.sh OUTDIR
#   full GPU suite, smoke, the driver's bench command, kernel traces (forward, backward), per-kernel HBM counters of the
#   same build, the other workloads' bench lines, the float64 mode, virtual ranks, the copy / row-pattern microbenchmark.
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$out/pytest.log"; tail -2 "$out/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$out/smoke.log"
tools/gpu_pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_kernels.json" "$out/pmc_kernels.json"; cp "$out/pmc/pmc_kernels.txt" "$out/pmc_kernels.txt"; cp "$out/pmc/kernel_stats.txt" "$out/kernel_stats_64k_sparse.txt"; rm -rf "$out/pmc"
cp "$out/pmc_kernels.json" profiles/r4_pmc_kernels.json   # bench.py reads the counter summary of THIS build from there
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_64k_sparse.json" 2> "$out/bench_64k_sparse.err"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$here/$out/kb" -o kb -- python "$here/tools/roundtrip_64k.py" > "$here/$out/kb.log" 2>&1 )
db=$(find "$out/kb" -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats_backward_64k_sparse.txt" 2>&1; rm -rf "$out/kb"
for w in 8k 12k 24k 32k-8x8 64k-sparse-4x4 128k 128k-8x8; do
  timeout 500 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --column-precision 64 > "$out/bench_64k_sparse_f64.json" 2> "$out/bench_64k_sparse_f64.err"
VR_WHOLE_WAVES=1 timeout 600 python tools/virtual_rank_time.py 64k-sparse "$out/virtual_ranks_64k-sparse_whole_waves.json" > "$out/virtual_ranks_whole.log" 2>&1
timeout 100 tools/mall_pipe.bin > "$out/mall_pipe.txt" 2>&1
python - "$out" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["ms_per_step"], "frac", d["roofline"]["frac"], "parity", (d.get("parity") or {}).get("rel_rmse"),
              "bwd", (d.get("backward") or {}).get("ms_per_pass"), (d.get("backward") or {}).get("parity", {}).get("rel_rmse"),
              "rt", (d.get("roundtrip") or {}).get("ms_per_pass"))
    except Exception as exc:
        print(f, "FAILED", exc)
PY
cut -c1-200 "$out/kernel_stats_64k_sparse.txt" | head -12
