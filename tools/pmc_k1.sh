#!/bin/bash
# PMC passes on the N=200 four-index transform (K1 contract_kernel): where do the SIMDs wait?
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/k1_pmc_a -o p -- python3 $R/tools/run_transform.py 200 2 > /dev/null 2> $R/gpurun_out/k1_pmc_a.err
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/k1_pmc_b -o p -- python3 $R/tools/run_transform.py 200 2 > /dev/null 2> $R/gpurun_out/k1_pmc_b.err
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --kernel-trace -d $R/gpurun_out/k1_pmc_c -o p -- python3 $R/tools/run_transform.py 200 2 > /dev/null 2> $R/gpurun_out/k1_pmc_c.err
echo ok1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/k1_pmc_d -o p -- python3 $R/tools/run_transform.py 200 2 > /dev/null 2> $R/gpurun_out/k1_pmc_d.err
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/k1_pmc_e -o p -- python3 $R/tools/run_transform.py 200 2 > /dev/null 2> $R/gpurun_out/k1_pmc_e.err
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/k1_trace -o p -- python3 $R/tools/run_transform.py 200 3 > /dev/null 2> $R/gpurun_out/k1_trace.err
echo ok2
