#!/bin/bash
# Issue-side counters of the packed stage-1 kernel on the bench command (how busy is the fp64 MFMA pipe,
# how many VALU instructions per MFMA), its HBM bytes, and the rocprofv3 --kernel-trace --stats summary
# of the same command.  Separate passes, each under its own timeout; summaries only are kept.
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--no-transform --no-berry --no-kupccd --no-cpu-baseline --steps 40 --warmup 5"
cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/s1_pmc_summary.json
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout -k 5 150 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $R/gpurun_out/s1i_pmc_$i -o p -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/s1i_pmc_$i.err
    echo "pass $i ($set): rc $?" | tee -a $R/gpurun_out/s1i_progress.log
    python3 $R/tools/pmc_stage1_summary.py $R/gpurun_out/s1i_pmc_$i | tail -1
    rm -rf $R/gpurun_out/s1i_pmc_$i
done
mv $R/gpurun_out/s1_pmc_summary.json $R/gpurun_out/s1_issue_counters.json
timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/s1i_stats -o p -- python3 $R/bench.py --no-transform --no-berry --no-kupccd --no-cpu-baseline --steps 200 > $R/gpurun_out/s1i_bench_under_rocprof.json 2> $R/gpurun_out/s1i_stats.err
echo "stats rc $?"
find $R/gpurun_out/s1i_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/s1i_kernel_stats.csv \;
rm -rf $R/gpurun_out/s1i_stats
