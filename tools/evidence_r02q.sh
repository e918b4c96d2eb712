#!/bin/bash
# Round-2 evidence run (one gpurun call): dependency probe, stage-1 cycle accounting, clocks and power,
# the bench on the driver's command, then tools/pmc_stage1_issue.sh (counters + rocprofv3 --kernel-trace --stats).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/q
mkdir -p $O
timeout -k 10 120 tools/bin/mfma_dep_probe > $O/mfma_dep_probe.txt || exit 1
echo "mfma probe done"
for p in 1 2 3; do echo "OOVQE_TRI_PROBE=$p"; timeout -k 10 60 tools/bin/tri_spread$p | tail -3 || exit 1; done > $O/stage1_cycles.txt
echo "stage-1 probes done"
timeout -k 10 200 python tools/clock_power_probe.py 256 > $O/clock_power.txt 2> $O/clock_power.err || exit 1
echo "clock probe done"
timeout -k 10 600 python bench.py --gpus 1 > $O/bench_full.json 2> $O/bench_full.err || exit 1
echo "bench done"
# issue counters, HBM bytes and the rocprofv3 --kernel-trace --stats summary (gpurun_out/s1_issue_counters.json,
# s1i_kernel_stats.csv, s1i_bench_under_rocprof.json)
bash tools/pmc_stage1_issue.sh || exit 1
echo "rocprof done"
