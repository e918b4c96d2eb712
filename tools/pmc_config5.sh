#!/bin/bash
# Issue-side counters of the config-5 kernels (kUpCCD CAS(8e,8o), batch 256: sector_rdm_fused_kernel and the lambda
# kernels of the adjoint) -- matrix-pipe busy cycles, VALU / LDS instruction activity, LDS bank conflicts -- and the
# rocprofv3 --kernel-trace --stats summary of the same command.  Separate passes, summaries only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/config5_pmc.txt
: > $OUT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
    i=$((i + 1))
    timeout -k 5 150 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $R/gpurun_out/c5_pmc_$i -o p -- python3 $R/tools/config5_batch.py 256 1 > /dev/null 2> $R/gpurun_out/c5_pmc_$i.err
    echo "pass $i ($set): rc $?" >> $OUT
    python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/c5_pmc_$i | grep -i "sector" >> $OUT
    rm -rf $R/gpurun_out/c5_pmc_$i $R/gpurun_out/c5_pmc_$i.err
done
timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/c5_stats -o p -- python3 $R/tools/config5_batch.py 256 1 > $R/gpurun_out/config5_batch_under_rocprof.txt 2> /dev/null
find $R/gpurun_out/c5_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/config5_kernel_stats.csv \;
rm -rf $R/gpurun_out/c5_stats
cat $OUT
