#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by several percent): tools/ab_bench.sh libdirA libdirB [rounds]
root=${GRAFT_REPO_ROOT:-/root/repo}
a=$1; b=$2; n=${3:-3}
for i in $(seq 1 $n); do
  for v in $a $b; do
    WOFT_HIP_LIB=$root/woft_amd/$v/libwoft_hip.so python $root/bench.py --no-alt-precisions --no-alt-corr --no-cpu-baseline --no-ladder --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'fps', round(d['ms_per_step'],3), 'ms; wh', round(d['roofline']['avg_launch_ms'],3))"
  done
done
