#!/bin/bash
# A/B of the update block's launch structure inside ONE gpurun call: tools/ab_pk.sh [rounds] [extra bench args]
# WOFT_UPDATE_PK=0: one launch per layer (round 3's 9 launches per iteration); 1: the persistent kernel (3 launches per iteration)
root=${GRAFT_REPO_ROOT:-/root/repo}
n=${1:-2}; shift
for i in $(seq 1 $n); do
  for v in 0 1; do
    WOFT_UPDATE_PK=$v python $root/bench.py --no-alt-precisions --no-alt-corr --no-cpu-baseline --no-ladder --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('UPDATE_PK=$v', round(d['value'],2), 'fps', round(d['ms_per_step'],3), 'ms; roofline', r['kernel'][:40], round(r['avg_launch_ms'],4), 'ms frac', round(r['frac'],4))"
  done
done
