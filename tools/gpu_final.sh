#!/bin/bash
# The round's final collection on the GPU box:  tools/gpu_final.sh OUTDIR
#   full GPU test suite, smoke, the driver's bench command (more steps), kernel traces (forward, backward), per-kernel
#   HBM counters, the other workloads' bench lines.  Copy OUTDIR/* to profiles/ under the round's prefix afterwards.
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$out/pytest.log"; tail -2 "$out/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$out/smoke.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_64k_sparse.json" 2> "$out/bench_64k_sparse.err"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$here/$out/kt" -o kt -- python "$here/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-backward > "$here/$out/kt.log" 2>&1 )
db=$(find "$out/kt" -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats_64k_sparse.txt" 2>&1; rm -rf "$out/kt"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$here/$out/kb" -o kb -- python "$here/tools/roundtrip_64k.py" > "$here/$out/kb.log" 2>&1 )
db=$(find "$out/kb" -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats_backward_64k_sparse.txt" 2>&1; rm -rf "$out/kb"
tools/gpu_pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_kernels.json" "$out/pmc_kernels.json"; cp "$out/pmc/pmc_kernels.txt" "$out/pmc_kernels.txt"; rm -rf "$out/pmc"
for w in 8k 12k 24k 32k-8x8 64k-sparse-4x4 128k 128k-8x8; do
  timeout 500 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
for w in 8k 24k 32k-8x8 128k; do tools/gpu_trace.sh "$out" $w > /dev/null 2>&1; done
# r4: the float64-arithmetic column passes as the timed configuration, its kernel trace, virtual ranks in both subgrid
# ownership modes, the K1 timeline, a functional 2-rank run of the multi-GPU bench path on this one GPU (gloo, host-staged)
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --column-precision 64 > "$out/bench_64k_sparse_f64.json" 2> "$out/bench_64k_sparse_f64.err"
BENCH_ARGS="--column-precision 64" tools/gpu_trace_only.sh "$out/trace_f64" A=1 > /dev/null 2>&1; cp "$out/trace_f64/kernel_stats.txt" "$out/kernel_stats_64k_sparse_f64.txt"; rm -rf "$out/trace_f64"
timeout 600 python tools/virtual_rank_time.py 64k-sparse "$out/virtual_ranks_64k-sparse.json" > "$out/virtual_ranks.log" 2>&1
VR_WHOLE_WAVES=1 timeout 600 python tools/virtual_rank_time.py 64k-sparse "$out/virtual_ranks_64k-sparse_whole_waves.json" > "$out/virtual_ranks_whole.log" 2>&1
SWIFTLY_HIP_LIB="$here/variants/trace.so" timeout 200 python tools/k1_trace.py > "$out/k1_trace.txt" 2>&1
SWIFTLY_BENCH_BACKEND=gloo SWIFTLY_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --no-backward 2>/dev/null | grep '^{' > "$out/bench_2rank_functional_gloo.json"
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
