#!/bin/bash
# One GPU-box session, parameterised (replaces the one-shot tools/gpu_s*.sh / gpu_final*.sh / gpu_iter.sh scripts of
# rounds 2-4).  Run through gpurun from the repository root:
#
#   gpurun -- 'tools/gpu_session.sh OUTNAME STEP [STEP ...]'        results under gpurun_out/OUTNAME/
#
# STEPs (executed in the order given; ENV=V words apply to the following steps' processes, `--` forgets them):
#   tests[:EXPR]         pytest -m gpu (-k EXPR), -x                     -> pytest.log
#   smoke                __graft_entry__.smoke()                          -> smoke.log
#   bench[:WORKLOAD]     bench.py --steps 10 --warmup 3 --no-cpu-baseline -> bench_WORKLOAD.json (+ one summary line)
#   driver               the driver's command: python bench.py            -> bench_driver.json
#   ab:NAME              the same short forward bench under the current ENV words -> bench_NAME.json (for A/B runs; use
#                        SWIFTLY_HIP_LIB=$PWD/variants/X.so to compare build variants, tools/build_variant.sh)
#   k1[:NAME]            tools/time_k1_band.py (K1 and the backward finish, one HIP-event pair per 9 launches) -> k1.txt
#   trace[:WORKLOAD]     rocprofv3 --kernel-trace of a 2-step forward bench -> kernel_stats_WORKLOAD.txt, timeline_WORKLOAD.csv,
#                        idle-gap accounting (tools/trace_timeline.py)
#   pmc:forward|backward per-kernel FETCH_SIZE / WRITE_SIZE + durations of this build (tools/gpu_pmc.sh) merged into
#                        pmc_kernels.json (copy it to profiles/r6_pmc_kernels.json)
#   others               bench lines of the other workloads (8k 12k 24k 32k-8x8 64k-sparse-4x4 128k 128k-8x8)
#   f64                  bench with --column-precision 64                  -> bench_64k_sparse_f64.json
#   vranks               tools/virtual_rank_time.py in both ownership modes
#   run:CMD              any command (quote it)                            -> appended to run.log
out=gpurun_out/$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
envs=()
summary() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "frac", d["hbm_algorithmic_frac_of_peak"], "parity",
          (d.get("parity") or {}).get("rel_rmse"), "bwd", (d.get("backward") or {}).get("ms_per_pass"),
          (d.get("backward") or {}).get("parity", {}).get("rel_rmse"), "rt", (d.get("roundtrip") or {}).get("ms_per_pass"),
          {k: v.get("total_ms", v.get("avg_ms")) for k, v in (d.get("stages") or {}).items()})
except Exception as exc:  # noqa
    print(sys.argv[1], "FAILED", exc)
PY
}
for step in "$@"; do
  if [[ "$step" =~ ^[A-Za-z_][A-Za-z0-9_]*= ]]; then envs+=("$step"); continue; fi   # ENV=V word
  if [ "$step" = "--" ]; then envs=(); continue; fi                                    # forget the ENV words
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case "$kind" in
    tests)
      if [ -n "$arg" ]; then env "${envs[@]}" timeout 1500 python -m pytest tests -m gpu -q -x -k "$arg" 2>&1 | tail -15 > "$out/pytest.log"
      else env "${envs[@]}" timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > "$out/pytest.log"; fi
      tail -3 "$out/pytest.log" ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$out/smoke.log" ;;
    bench)
      wl=${arg:-64k-sparse}
      env "${envs[@]}" timeout 600 python bench.py --workload "$wl" --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/bench_$wl.json" 2> "$out/bench_$wl.err"
      summary "$out/bench_$wl.json" ;;
    driver)
      env "${envs[@]}" timeout 900 python bench.py > "$out/bench_driver.json" 2> "$out/bench_driver.err"; summary "$out/bench_driver.json" ;;
    ab)
      env "${envs[@]}" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-backward --no-other-workloads ${BENCH_ARGS:-} > "$out/bench_$arg.json" 2> "$out/bench_$arg.err"
      summary "$out/bench_$arg.json" ;;
    k1)
      env "${envs[@]}" timeout 300 python tools/time_k1_band.py 2>&1 | grep "ms per facet" | sed "s/^/${arg:-default}: /" | tee -a "$out/k1.txt" ;;
    trace)
      wl=${arg:-64k-sparse}
      ( cd /tmp && env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d "$here/$out/kt" -o kt -- python "$here/bench.py" --workload "$wl" --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-backward ${BENCH_ARGS:-} > "$here/$out/kt.log" 2>&1 )
      db=$(find "$out/kt" -name '*.db' | head -1)
      if [ -n "$db" ]; then
        python tools/rocpd_stats.py "$db" > "$out/kernel_stats_$wl.txt" 2>&1
        python tools/trace_timeline.py dump "$db" "$out/timeline_$wl.csv" > /dev/null
        python tools/trace_timeline.py gaps "$out/timeline_$wl.csv" > "$out/timeline_$wl.txt"; head -12 "$out/timeline_$wl.txt"
      fi
      rm -rf "$out/kt" ;;
    pmc)
      tools/gpu_pmc.sh "$out/pmc" 64k-sparse "${arg:-forward}" > "$out/pmc_${arg:-forward}.log" 2>&1
      cp "$out/pmc/pmc_kernels.json" "$out/pmc_kernels.json"; cp "$out/pmc/pmc_kernels.json" profiles/r6_pmc_kernels.json
      cp "$out/pmc/kernel_stats_${arg:-forward}.txt" "$out/kernel_stats_serial_${arg:-forward}.txt"
      cat "$out/pmc/pmc_kernels_${arg:-forward}.txt" ;;
    others)
      for w in 8k 12k 24k 32k-8x8 64k-sparse-4x4 128k 128k-8x8; do
        env "${envs[@]}" timeout 500 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
        summary "$out/bench_$w.json"
      done ;;
    f64)
      timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --column-precision 64 > "$out/bench_64k_sparse_f64.json" 2> "$out/bench_64k_sparse_f64.err"
      summary "$out/bench_64k_sparse_f64.json" ;;
    vranks)
      timeout 600 python tools/virtual_rank_time.py 64k-sparse "$out/virtual_ranks_64k-sparse.json" > "$out/virtual_ranks.log" 2>&1; tail -4 "$out/virtual_ranks.log"
      VR_WHOLE_WAVES=1 timeout 600 python tools/virtual_rank_time.py 64k-sparse "$out/virtual_ranks_64k-sparse_whole_waves.json" > "$out/virtual_ranks_whole.log" 2>&1; tail -4 "$out/virtual_ranks_whole.log" ;;
    run) ( env "${envs[@]}" timeout 900 bash -c "$arg" ) 2>&1 | tee -a "$out/run.log" | tail -40 ;;
    *) echo "unknown step: $step" ;;
  esac
done
