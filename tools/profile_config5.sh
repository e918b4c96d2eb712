#!/bin/bash
# rocprofv3 kernel stats of the configs[4] workload (kUpCCD CAS(8e,8o), sector engine).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${1:-rXX}_c5 -o c5 -- python3 $R/tools/${2:-config5_breakdown.py} > $R/gpurun_out/${1:-rXX}_c5.log 2> $R/gpurun_out/${1:-rXX}_c5.err
echo profiled
