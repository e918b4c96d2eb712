#!/bin/bash
# HBM traffic of eri_ingest_kernel (rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE: separate passes) and its kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ingest_pmc.txt
: > $OUT
python3 $R/tools/ingest_probe.py 256 2>/dev/null >> $OUT
for set in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout -k 5 200 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $R/gpurun_out/ing_pmc -o p -- python3 $R/tools/ingest_probe.py 256 > /dev/null 2> /dev/null
    echo "pass $set: rc $?" >> $OUT
    python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/ing_pmc | grep -i "eri_ingest\|eri_pack" >> $OUT
    rm -rf $R/gpurun_out/ing_pmc
done
timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/ing_stats -o p -- python3 $R/tools/ingest_probe.py 256 > /dev/null 2> /dev/null
find $R/gpurun_out/ing_stats -name "*kernel_stats.csv" -exec grep -i "eri_ingest\|eri_pack\|Name" {} \; | cut -c1-200 >> $OUT
rm -rf $R/gpurun_out/ing_stats
cat $OUT
