"""Independent batched evaluations issued alternately on two HIP streams (workspace slots 0 / 1): the stage-1 sweep
of one call keeps one workgroup per geometry busy for ~380 us and fills their CUs' register files; with fewer than
256 geometries per call the remaining CUs are free for the latency-bound tail kernels of the OTHER stream's call.
    python tools/two_stream_probe.py [G ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 240, 224, 208, 192, 176, 160, 128]
pqc, batch, single, thetas = bench.build_geometries(list(range(max(sizes))))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(G, n_calls, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if two:
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for i in range(n_calls):
            with torch.cuda.stream(streams[i & 1]):
                batch.evaluate(thetas, count=G, slot=i & 1)
    else:
        for i in range(n_calls):
            batch.evaluate(thetas, count=G)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_calls


for G in sizes:
    for two in (False, True):
        run(G, 6, two)
    t1 = min(run(G, 60, False) for _ in range(3))
    t2 = min(run(G, 60, True) for _ in range(3))
    print(f"G = {G:3d}: one stream {t1 * 1e6:7.1f} us per call = {G / t1 / 1e3:6.1f}K evaluations/s; "
          f"two streams {t2 * 1e6:7.1f} us = {G / t2 / 1e3:6.1f}K", flush=True)
