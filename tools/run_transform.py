#!/usr/bin/env python3
"""Run the full 4-index transform a few times (for rocprofv3 counter passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_oo_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.rand((N, N, N, N), dtype=torch.float64, device="cuda") - 0.5
C = torch.rand((N, N), dtype=torch.float64, device="cuda") - 0.5
o = torch.empty_like(g); w = torch.empty_like(g)
for _ in range(reps):
    ops.general_4index_transform(g, C, C, C, C, out=o, work=w)
torch.cuda.synchronize()
print("done")
