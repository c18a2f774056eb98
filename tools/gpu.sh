#!/bin/bash
# Rebuild the HIP library in-tree, then run a command on the MI355X box via gpurun.
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python adaptive-depth-u-net-for-image-super-resolution-segmentation_amd/build.py > /tmp/adunet_build.log 2>&1 || { cat /tmp/adunet_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
