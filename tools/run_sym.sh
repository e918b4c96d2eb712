set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "symmetr or eri_pack" > gpurun_out/t_sym.log 2>&1
timeout -k 10 900 python -m pytest tests/test_api_gpu.py tests/test_full_size_gpu.py -q -x -k "batch" > gpurun_out/t_sym2.log 2>&1
