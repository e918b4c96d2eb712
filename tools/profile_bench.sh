#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats + PMC passes (separate runs) of the bench command.
# usage: bash tools/profile_bench.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--no-transform --no-berry --no-kupccd --no-cpu-baseline --steps 200 --warmup 20"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace -o trace -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_trace.json 2> $R/gpurun_out/${TAG}_trace.err
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_fetch -o pmc -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_write -o pmc -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
echo profiled
