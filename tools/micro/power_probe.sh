#!/bin/bash
# Socket power / shader clock while the tracker runs: tools/micro/power_probe.sh  (on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-alt-precisions --no-alt-corr "$@" > /tmp/b.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo
    sleep 0.5
done | sort | uniq -c | sort -k1,1nr | head -12
python -c "
import json;d=json.loads(open('/tmp/b.json').read().strip().split(chr(10))[-1]);print('fps',round(d['value'],1))"
