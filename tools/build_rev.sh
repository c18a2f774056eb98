#!/bin/bash
# Build the HIP library of a given git revision into ab/<name>.so (for same-box A/B timing through ADUNET_LIB).
# usage: tools/build_rev.sh <rev> <name>
set -e
cd "$(dirname "$0")/.."
rev=$1; name=$2
pkg=adaptive-depth-u-net-for-image-super-resolution-segmentation_amd
tmp=$(mktemp -d)
mkdir -p "$tmp/$pkg/csrc" "$tmp/include" ab
git show "$rev:include/adunet.h" > "$tmp/include/adunet.h"
for f in $(git ls-tree --name-only "$rev" "$pkg/csrc/"); do git show "$rev:$f" > "$tmp/$f"; done
objs=""
for src in "$tmp/$pkg"/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -c "$src" -o "${src%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "ab/$name.so" "$tmp/$pkg"/csrc/*.o -ldl
rm -rf "$tmp"
echo "ab/$name.so"
