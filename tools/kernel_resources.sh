#!/bin/bash
# Static resource usage (VGPRs, scratch bytes/lane, occupancy) of every kernel in one .hip file, from the compiler's
# kernel-resource-usage remarks.  Usage: tools/kernel_resources.sh csrc/row_pass.hip [extra hipcc flags]
src=$(readlink -f "$1"); shift
tmp=$(mktemp -d)
(cd "$tmp" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-slp-vectorize "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o "$tmp/o.o" 2>&1) |
    sed 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
    awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {printf "%4s vgpr %5s scratch  occ %s  %s\n", v, s, o, name}' | sort -k7
rm -rf "$tmp"
