#!/bin/bash
# Build the working tree's HIP library with extra compiler flags into ab/<name>.so (diagnostic variants for ADUNET_LIB).
# usage: tools/build_variant.sh <name> <flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
pkg=adaptive-depth-u-net-for-image-super-resolution-segmentation_amd
tmp=$(mktemp -d)
mkdir -p ab
for src in $pkg/csrc/*.hip; do
  b=$(basename "${src%.hip}")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value "$@" -c "$src" -o "$tmp/$b.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "ab/$name.so" "$tmp"/*.o -ldl
rm -rf "$tmp"
echo "ab/$name.so"
