#!/bin/bash
# PMC passes over one conv launch variant: tools/pmc_conv.sh <tag> <case> <halo> [tile_n] -> gpurun_out/pmcc_<tag>.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmcc_$tag.txt
: > $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    rm -rf /tmp/pmcc_$tag
    (cd $root && rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcc_$tag -o p -- python tools/regb_probe.py "$@") > /tmp/pmcc_$tag.log 2>&1
    f=$(find /tmp/pmcc_$tag -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python $root/tools/show_pmc.py "$f" --match conv_ >> $out; else echo "FAILED: $grp" >> $out; tail -3 /tmp/pmcc_$tag.log >> $out; fi
done
