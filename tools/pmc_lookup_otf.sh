#!/bin/bash
# PMC passes over the volume-free lookup alone (tools/bench_lookup_otf.py) -> gpurun_out/pmcc_lookup_otf.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmcc_lookup_otf.txt
: > $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    rm -rf /tmp/pmcl
    (cd $root && rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcl -o p -- python tools/bench_lookup_otf.py) > /tmp/pmcl.log 2>&1
    f=$(find /tmp/pmcl -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python $root/tools/show_pmc.py "$f" --match corr_lookup_otf >> $out; else echo "FAILED: $grp" >> $out; tail -3 /tmp/pmcl.log >> $out; fi
done
