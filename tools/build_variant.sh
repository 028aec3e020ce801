#!/bin/bash
# Build a variant of libcenternet_amd.so with extra compiler flags (A/B of build-time choices):
#   tools/build_variant.sh <name> "<extra hipcc flags>"   ->  centernet_amd/variants/libcenternet_amd_<name>.so
# Use with CENTERNET_AMD_LIB=<that path>.  Objects go to /tmp; the in-tree build is untouched.
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/centernet_amd/csrc
obj=/tmp/cn_variant_$name
mkdir -p "$obj" "$root/centernet_amd/variants"
objs=""
for f in $(sed -n "s/^SRCS *:= *//p" "$src/Makefile" | sed "s/\.hip//g"); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra -c "$src/$f.hip" -o "$obj/$f.o" &
  objs="$objs $obj/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/centernet_amd/variants/libcenternet_amd_$name.so" $objs
echo "$root/centernet_amd/variants/libcenternet_amd_$name.so"
