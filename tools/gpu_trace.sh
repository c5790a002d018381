#!/bin/bash
# rocprofv3 kernel trace of one forward pass of a workload:  tools/gpu_trace.sh OUTDIR WORKLOAD  -> OUTDIR/kernel_stats_WORKLOAD.txt
out=$1; wl=$2
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$here/$out/kt_$wl" -o kt -- python "$here/bench.py" --workload "$wl" --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-backward > "$here/$out/kt_$wl.log" 2>&1 )
db=$(find "$out/kt_$wl" -name '*.db' | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$out/kernel_stats_$wl.txt" 2>&1
rm -rf "$out/kt_$wl"
cut -c1-210 "$out/kernel_stats_$wl.txt" | head -14
