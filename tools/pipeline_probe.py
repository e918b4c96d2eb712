"""Independent evaluation calls: in order on the current stream, and deferred over the library's two side streams
(OO_pqc_batch.evaluate_deferred), N^4 sweeps of different streams ordered (default) or free-running.
Prints us per call, evaluations/s, the HIP-event duration of the sweep, bit-identity."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from auto_oo_amd import ops, _lib  # noqa: E402


def main():
    geoms = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    pqc, batch, single, thetas = bench.build_geometries(list(range(geoms)))
    ref = batch.evaluate(thetas).clone()

    def run(n, count, deferred):
        torch.cuda.synchronize()
        ops.profile_begin()
        t = time.perf_counter()
        outs = []
        for i in range(n):
            outs.append(batch.evaluate_deferred(thetas, count=count) if deferred else batch.evaluate(thetas, count=count))
        outs = [o.result() if deferred else o for o in outs]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        ms, cnt, _ = ops.profile_end()
        same = all(torch.equal(o, ref[:count]) for o in outs[-3:])
        return dt, ms / max(cnt, 1) * 1e3, same

    for count in (geoms, geoms - 32):
        for free in (0, 1):
            for deferred in (False, True):
                if free and not deferred:
                    continue
                lib.oovqe_debug_set_option(b"stage1_free_run", free)
                run(8, count, deferred)
                best = min(run(48, count, deferred) for _ in range(3))
                print(f"G={count:4d} sweeps_ordered={1 - free} "
                      f"deferred={int(deferred)}: {best[0] * 1e6:8.1f} us/call {count / best[0]:10.0f} evals/s "
                      f"sweep {best[1]:7.1f} us bit-identical={best[2]}", flush=True)
    lib.oovqe_debug_set_option(b"stage1_free_run", 0)


if __name__ == "__main__":
    main()
