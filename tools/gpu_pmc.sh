#!/bin/bash
# Per-kernel HBM traffic of the forward pass (three rocprofv3 runs of the same command; counters in their own runs):
#   tools/gpu_pmc.sh OUTDIR [workload] [forward|backward]   ->  OUTDIR/pmc_kernels.json (merge into profiles/r6_pmc_kernels.json;
#   carries the build identity: source hash, .so hash, git commit); `backward` traces tools/run_backward.py (band schedule)
out=$1; wl=${2:-64k-sparse}; dir=${3:-forward}
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
# (r4 session 3) every kernel ALONE on the chip: the planned-wave prefetch and the two-stream K2 chunks are switched off for
# the collection, so that a launch is one kernel of one wave and its duration is not shared with a concurrent kernel; the
# production pass overlaps them (bench.py's ms_per_step) -- byte counts per kernel do not depend on the schedule
export SWIFTLY_PREFETCH=0 SWIFTLY_K2_CHUNK=0
cmd="python $here/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-backward"
[ "$dir" = backward ] && cmd="python $here/tools/run_backward.py $wl 2"
[ -f "$here/profiles/r6_pmc_kernels.json" ] && cp "$here/profiles/r6_pmc_kernels.json" "$out/pmc_kernels.json"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$here/$out/kt" -o kt -- $cmd > "$here/$out/kt.log" 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$here/$out/pf" -o pf -- $cmd > "$here/$out/pf.log" 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$here/$out/pw" -o pw -- $cmd > "$here/$out/pw.log" 2>&1 )
kt=$(find "$out/kt" -name '*.db' | head -1); pf=$(find "$out/pf" -name '*.db' | head -1); pw=$(find "$out/pw" -name '*.db' | head -1)
python tools/rocpd_stats.py "$kt" > "$out/kernel_stats_$dir.txt" 2>&1
python tools/pmc_kernels.py "$kt" "$pf" "$pw" "$out/pmc_kernels.json" "$wl" "$dir" | tee "$out/pmc_kernels_$dir.txt"
rm -rf "$out/kt" "$out/pf" "$out/pw"
