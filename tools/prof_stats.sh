#!/bin/bash
# Per-kernel time table of one bench.py run on the GPU box (run through gpurun):
#   tools/prof_stats.sh <tag> [bench.py args...]   ->  gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- \
    python $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ladder --no-alt-precisions --no-alt-corr "$@" > /tmp/prof_$tag.log 2>&1
mkdir -p $root/gpurun_out
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv && tail -1 /tmp/prof_$tag.log | cut -c1-200
