#!/bin/bash
# rocprofv3 kernel trace of a 2-step forward-only bench: tools/gpu_trace_only.sh OUTDIR [ENV=V ...]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace -d "$OLDPWD/$out/kt" -o kt -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-backward ${BENCH_ARGS:-} > "$OLDPWD/$out/kt.log" 2>&1 )
db=$(find "$out/kt" -name '*.db' | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats.txt" 2>&1
rm -rf "$out/kt"
cut -c1-230 "$out/kernel_stats.txt" | head -14
