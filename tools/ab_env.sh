#!/bin/bash
# Same-box A/B of an environment switch: alternates bench.py runs with VAR=<off> / VAR=<on> and prints ms/step.
# usage (on the GPU box): tools/ab_env.sh <rounds> VAR [off-value on-value [bench.py arguments ...]]
#   tools/ab_env.sh 3 ADUNET_KEEP_HEAD_ACT                      (unset / 1, the K2' step)
#   tools/ab_env.sh 3 ADUNET_NO_PW_WIDE 1 '' --workload E2s06       (first value = the baseline, '' = unset)
rounds=$1; var=$2; off=${3:-}; on=${4:-1}
shift 2; [ $# -ge 2 ] && shift 2
for r in $(seq "$rounds"); do
  for v in off on; do
    val=$off; [ "$v" = on ] && val=$on
    if [ -n "$val" ]; then export "$var"="$val"; else unset "$var"; fi
    ms=$(python bench.py --steps 40 --warmup 8 --no-micro --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $var=$v  $ms ms/step"
  done
done
