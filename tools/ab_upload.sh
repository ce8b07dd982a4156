#!/bin/bash
# host-frame upload path A/B inside ONE gpurun call: the runtime's pageable copy (default) against the pinned staging path in 4 pieces
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in direct staged; do
    WOFT_UPLOAD=$v python $root/bench.py --no-alt-precisions --no-alt-corr --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('WOFT_UPLOAD=$v resident', round(d['value'],2), 'host frames', round(d['host_frames']['frames_per_s'],2), 'ref-form', d['config']['fps_reference_form_config_unmodified'], d['config']['fps_reference_form_config_fp32'])"
  done
done
