#!/bin/bash
# CPU CONTAINER ONLY (listed in .gpurunignore: never sent to a GPU box).  Builds the C-ABI library with AddressSanitizer +
# UndefinedBehaviorSanitizer on its HOST code only (-Xarch_host: the gfx950 code objects are the ordinary, uninstrumented ones;
# GPU sanitizer builds are not available on the pool and are not what this is) for the host-arithmetic fuzz of
# tests/test_host_sanitizers.py / tools/host_sanitizer_fuzz.py.
# usage: tools/build_host_sanitized.sh <output directory>      -> <output directory>/libadunet_san.so
set -e
out=$1
root="$(cd "$(dirname "$0")/.." && pwd)"
csrc="$root/adaptive-depth-u-net-for-image-super-resolution-segmentation_amd/csrc"
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$out"
for src in "$csrc"/*.hip; do
  b=$(basename "${src%.hip}")
  "$hipcc" --offload-arch=gfx950 -O1 -g1 -fPIC -std=c++17 -Wno-unused-value -Xarch_host -fsanitize=address,undefined \
      -Xarch_host -fno-sanitize-recover=undefined -c "$src" -o "$out/$b.o" &
done
wait
"$hipcc" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$out/libadunet_san.so" "$out"/*.o -ldl
echo "$out/libadunet_san.so"
