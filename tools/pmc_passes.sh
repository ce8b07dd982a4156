#!/bin/bash
# PMC counter passes (one rocprofv3 run per group; counters only, no tracing) over a command, on the GPU box:
#   tools/pmc_passes.sh <tag> '<command>'    ->  gpurun_out/pmc_<tag>_<pass>.csv
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA"; do
    rm -rf /tmp/pmc_$tag
    (cd $root && rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -o p -- bash -c "$*") > /tmp/pmc_$tag.log 2>&1
    f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp "$f" $root/gpurun_out/pmc_${tag}_$i.csv || tail -5 /tmp/pmc_$tag.log
    i=$((i+1))
done
ls -la $root/gpurun_out/ | grep pmc_$tag
