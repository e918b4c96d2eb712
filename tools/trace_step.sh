#!/bin/bash
# rocprofv3 kernel trace of lockstep Newton steps over G geometries; lists the launches of one step (tools/trace_one_call.py).
#   tools/trace_step.sh G out.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
G=${1:-8}
OUT=${2:-gpurun_out/step_trace_G$G.txt}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/_trace_tmp
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d $R/gpurun_out/_trace_tmp -o p -- python3 $R/tools/lockstep_trace.py $G > /dev/null 2> $R/gpurun_out/_trace_tmp.err
f=$(find $R/gpurun_out/_trace_tmp -name '*kernel_trace.csv' | head -1)
python3 $R/tools/trace_one_call.py $f linesearch_update > $OUT
rm -rf $R/gpurun_out/_trace_tmp $R/gpurun_out/_trace_tmp.err
cat $OUT
