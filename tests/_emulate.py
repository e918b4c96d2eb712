"""numpy emulation of the gate-table semantics of include/oovqe.h (test helper only)."""
import numpy as np


def apply_gate_table(gates, theta, n, init_index):
    D = 1 << n
    psi = np.zeros(D)
    psi[init_index] = 1.0
    x = np.arange(D, dtype=np.uint32)
    for g in gates:
        if g.theta_idx < 0:
            continue
        fm = g.mask_hi | g.mask_lo
        sel = x[(x & fm) == g.mask_hi]
        y = sel ^ fm
        par = np.array([bin(int(v) & g.mask_par).count("1") & 1 for v in sel])
        pi = np.where(par == 1, -1.0, 1.0)
        a = 0.5 * g.sign * theta[g.theta_idx]
        c, s = np.cos(a), np.sin(a)
        ax, ay = psi[sel].copy(), psi[y].copy()
        psi[sel] = c * ax + pi * s * ay
        psi[y] = c * ay - pi * s * ax
    return psi
