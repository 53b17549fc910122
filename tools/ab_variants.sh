#!/bin/bash
# Builds variants of the library that differ in compile-time switches of ONE source file, HERE (hipcc cross-compiles),
# so that a single gpurun call can time them side by side on one box:
#   tools/ab_variants.sh collapse_lds.hip  v512:"-DFQD_FC_SLOTS=512 -DFQD_FC_WAVES=6"  w8:"-DFQD_FC_WAVES=8"
# -> fastqdedup_amd/libfqdedup_hip.v512.so, ...w8.so (git-ignored; selected with FQD_LIB_VARIANT=v512)
set -e
cd "$(dirname "$0")/.."
python -m fastqdedup_amd.build > /dev/null
src=$1; shift
obj=fastqdedup_amd/build/${src%.hip}.o
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c fastqdedup_amd/csrc/$src -o /tmp/ab_$name.o
    objs=$(ls fastqdedup_amd/build/*.o | grep -v "$obj")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ab_$name.o -o fastqdedup_amd/libfqdedup_hip.$name.so
    echo "built variant $name ($flags)"
done
