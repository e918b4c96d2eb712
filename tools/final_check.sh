#!/bin/bash
# Round-end check on the GPU box: full GPU tests, smoke(), default bench, 2-rank gloo rehearsal.
set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1
tail -n 2 gpurun_out/final_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1
tail -n 1 gpurun_out/final_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python -c "import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['roofline']['frac'], d['cpu_baseline'], d['transform']['tflops'], d['berry_loop'])"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --geoms 16 --steps 1600 --warmup 160 --no-transform --no-berry > gpurun_out/final_bench_gloo2.json 2> gpurun_out/final_bench_gloo2.err
tail -c 400 gpurun_out/final_bench_gloo2.json
