#!/bin/bash
# Round-2 evidence run (one gpurun call): dependency probe, stage-1 cycle accounting, clocks and power,
# the bench on the driver's command and the same under rocprofv3 --kernel-trace --stats.
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
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02q -- python3 bench.py --steps 200 --no-transform --no-berry --no-kupccd --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
echo "rocprof done"
