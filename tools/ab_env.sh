#!/bin/bash
# Same-box A/B of an environment switch: alternates bench.py runs without / with VAR=1 and prints ms/step.
# usage (on the GPU box): tools/ab_env.sh <rounds> VAR
rounds=$1; var=$2
for r in $(seq "$rounds"); do
  for on in 0 1; do
    if [ "$on" = 1 ]; then export "$var"=1; else unset "$var"; fi
    ms=$(python bench.py --steps 60 --warmup 10 --no-micro --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $var=$on  $ms ms/step"
  done
done
