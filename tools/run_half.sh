set -e
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "half_transform_sizes or cas_pipeline or cas_eval_fused" > gpurun_out/t1.log 2>&1


timeout -k 10 300 python -m pytest tests/test_api_gpu.py -q -x -k "batched" > gpurun_out/t4.log 2>&1
timeout -k 10 60 tools/bin/hs_base > gpurun_out/hs.log 2>&1
timeout -k 10 300 python bench.py --no-transform --no-berry --no-cpu-baseline > gpurun_out/b_fused.json 2>/dev/null
