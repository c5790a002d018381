#!/bin/bash
# SQ counters of K1 alone (tools/time_k1_only.py) in two rocprofv3 passes: tools/gpu_sq_pmc.sh OUTDIR [ENV=V ...]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
here=$(pwd)
for w in "$@"; do export "$w"; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d "$here/$out/p1" -o p1 -- python $here/tools/time_k1_only.py > "$here/$out/p1.log" 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU -d "$here/$out/p2" -o p2 -- python $here/tools/time_k1_only.py > "$here/$out/p2.log" 2>&1 )
for p in p1 p2; do db=$(find "$out/$p" -name '*.db' | head -1); [ -n "$db" ] && python tools/pmc_stats.py "$db" | grep -A12 "row_pass\|whole" ; done > "$out/sq.txt" 2>&1
tail -5 "$out/p1.log" >> "$out/sq.txt"
rm -rf "$out/p1" "$out/p2"
