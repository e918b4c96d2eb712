set -e
timeout -k 10 60 tools/bin/hs_base > gpurun_out/hs_sym.log 2>&1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "pq_symmetr or streaming or beyond_fused or half_transform_sizes" > gpurun_out/t_sym.log 2>&1
