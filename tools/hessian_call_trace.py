"""One energy_gradient_hessian call of G geometries, 12 times, for rocprofv3 --kernel-trace (tools/trace_one_call.py
lists the launches of one call: marker = hess_matrix_kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pqc, batch, single, thetas = bench.build_geometries(list(range(G)))
for _ in range(12):
    out = batch.energy_gradient_hessian(thetas)
torch.cuda.synchronize()
