#!/bin/bash
# A/B of an environment switch inside ONE gpurun call: tools/ab_env.sh VAR valueA valueB [rounds] [extra bench args]
root=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; a=$2; b=$3; n=${4:-3}; shift 4
for i in $(seq 1 $n); do
  for v in $a $b; do
    env $var=$v python $root/bench.py --no-alt-precisions --no-alt-corr --no-cpu-baseline --no-ladder --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['value'],2), 'fps', round(d['ms_per_step'],3), 'ms; wh', round(d['roofline']['avg_launch_ms'],3))"
  done
done
