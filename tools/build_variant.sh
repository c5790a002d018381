#!/bin/bash
# Build a variant of libswiftly_hip.so for same-box A/B runs (selected at run time with SWIFTLY_HIP_LIB):
#   tools/build_variant.sh NAME "file1.hip file2.hip" "-DFLAG=1 ..."
# recompiles the listed translation units with the extra flags and links them with the objects of the default build
# into variants/NAME.so (git-ignored, travels to the GPU box).
set -e
name=$1; files=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/ska-sdp-distributed-fourier-transform_amd/csrc
make -C "$csrc" -j8 > /dev/null
mkdir -p "$root/variants" "$csrc/build/var_$name"
objs=""
for o in "$csrc"/build/*.o; do
  b=$(basename "$o" .o)
  if echo " $files " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-slp-vectorize -Wall -Wno-unused-function $flags -c "$csrc/$b.hip" -o "$csrc/build/var_$name/$b.o" &
    objs="$objs $csrc/build/var_$name/$b.o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/variants/$name.so" $objs
echo "built variants/$name.so"
