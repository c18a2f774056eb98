#!/bin/bash
# Same-box A/B of two builds of the library: alternates bench.py runs (graph-replayed K2' step) and prints ms/step.
# usage (on the GPU box): tools/ab.sh <rounds> ab/base.so [ab/other.so ...]   ("-" = the in-tree library)
rounds=$1; shift
for r in $(seq "$rounds"); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset ADUNET_LIB; else export ADUNET_LIB="$PWD/$lib"; fi
    ms=$(python bench.py --steps 60 --warmup 10 --no-micro --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $lib  $ms ms/step"
  done
done
