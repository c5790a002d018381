#!/bin/bash
# HBM counters of the axis-1-first order (bench.py --axis1-first, fused K1): FETCH_SIZE and WRITE_SIZE in their own rocprofv3
# runs, every kernel alone on the chip (prefetch and K2 chunk streams off), per-kernel averages by tools/pmc_stats.py
#   gpurun -- 'bash tools/gpu_pmc_accurate.sh'   ->  gpurun_out/r6acc/pmc_accurate.txt (copy the swf kernels' rows to
#   profiles/r6_accurate_pmc_kernels.txt)
export TMPDIR=/tmp; here=$(pwd); o=$here/gpurun_out/r6acc; mkdir -p $o
export SWIFTLY_PREFETCH=0 SWIFTLY_K2_CHUNK=0
cmd="python $here/bench.py --axis1-first --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-backward --no-other-workloads"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $o/pf -o pf -- $cmd > $o/pf.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/pw -o pw -- $cmd > $o/pw.log 2>&1 )
python tools/pmc_stats.py $(find $o/pf -name '*.db' | head -1) $(find $o/pw -name '*.db' | head -1) > $o/pmc_accurate.txt 2>&1
rm -rf $o/pf $o/pw
unset SWIFTLY_PREFETCH SWIFTLY_K2_CHUNK
