#!/bin/bash
# Memory-system counters of the packed stage-1 kernel (half_tri_kernel) on the bench command: is the
# read stream held back inside the CU (TCP stalls), in the L2 (tag / FIFO stalls) or behind it (DRAM
# credit stalls of the L2's external-access unit, average read latency)?  Separate rocprofv3 passes of
# at most three counters (more TCC counters in one pass: "exceeds the capabilities of the hardware"),
# each under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--no-transform --no-berry --no-kupccd --no-cpu-baseline --steps 40 --warmup 5"
cd /tmp && export TMPDIR=/tmp
i=0
SETS_FROM=${1:-1}
for set in "TCC_CYCLE TCC_BUSY TCC_EA0_RDREQ" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_32B" \
           "TCC_REQ TCC_HIT TCC_MISS" "TCC_TAG_STALL TCC_LATENCY_FIFO_FULL TCC_SRC_FIFO_FULL" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" \
           "SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES"; do
    i=$((i + 1))
    if [ $i -lt $SETS_FROM ]; then continue; fi
    timeout -k 5 150 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $R/gpurun_out/s1_pmc_$i -o p -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/s1_pmc_$i.err
    echo "pass $i ($set): rc $?" | tee -a $R/gpurun_out/s1_pmc_progress.log
    python3 $R/tools/pmc_stage1_summary.py $R/gpurun_out/s1_pmc_$i | tail -1
    rm -rf $R/gpurun_out/s1_pmc_$i
done
