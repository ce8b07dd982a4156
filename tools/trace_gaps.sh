#!/bin/bash
# Kernel trace of a short bench run on the GPU box -> gpurun_out/<tag>_kernel_trace.csv (start/end per dispatch)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o $tag -- \
    python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ladder --no-alt-precisions --no-alt-corr "$@" > /tmp/tr_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name '*kernel_trace.csv' | head -1)
mkdir -p $root/gpurun_out && cp "$f" $root/gpurun_out/${tag}_kernel_trace.csv && wc -l $root/gpurun_out/${tag}_kernel_trace.csv
