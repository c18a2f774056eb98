#!/bin/bash
# Same-box A/B of an environment switch: alternates bench.py runs with VAR=<off> / VAR=<on> and prints ms/step.
# usage (on the GPU box): tools/ab_env.sh <rounds> VAR [off-value on-value [bench.py arguments ...]]
#   tools/ab_env.sh 3 ADUNET_KEEP_HEAD_ACT                      (unset / 1, the K2' step)
#   tools/ab_env.sh 3 AD_PW_NW8 0 1 --workload E2s06
rounds=$1; var=$2; off=${3:-}; on=${4:-1}
shift 2; [ $# -ge 2 ] && shift 2
for r in $(seq "$rounds"); do
  for v in off on; do
    if [ "$v" = on ]; then export "$var"="$on"; elif [ -n "$off" ]; then export "$var"="$off"; else unset "$var"; fi
    ms=$(python bench.py --steps 40 --warmup 8 --no-micro --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $var=$v  $ms ms/step"
  done
done
