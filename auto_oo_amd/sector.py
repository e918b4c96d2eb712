"""Particle-number-sector statevector engine (host side): string tables + thin wrappers over
``oovqe_sector_*`` (include/oovqe.h).  Used by ``Parameterized_circuit`` for active spaces whose
2^n register does not fit the one-workgroup LDS kernel (n_qubits > 10), e.g. kUpCCD CAS(8e,8o)."""
import ctypes
from itertools import combinations

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, stream_ptr

F64 = torch.float64


def string_tables(ncas, n_occ_strings):
    """Occupation strings of ``n_occ_strings`` particles in ``ncas`` orbitals (orbital p at bit
    ncas-1-p), ascending by value, and the inverse map (string -> index, -1 elsewhere)."""
    strings = []
    for occ in combinations(range(ncas), n_occ_strings):
        m = 0
        for p in occ:
            m |= 1 << (ncas - 1 - p)
        strings.append(m)
    strings.sort()
    unrank = np.array(strings, dtype=np.uint32)
    rank = np.full(1 << ncas, -1, dtype=np.int32)
    rank[unrank] = np.arange(len(strings), dtype=np.int32)
    return unrank, rank


def sector_of(hfstate, ncas):
    """(N_alpha, N_beta) of an occupation vector in interleaved ordering (even wire = alpha)."""
    occ = np.asarray(hfstate, dtype=int)
    return int(occ[0::2].sum()), int(occ[1::2].sum())


class SectorEngine:
    def __init__(self, ncas, hfstate, gates_dev, n_gates, n_theta, init_index, device):
        self.lib = _lib.load()
        self.ncas, self.device = ncas, device
        self.n_alpha, self.n_beta = sector_of(hfstate, ncas)
        ua, ra = string_tables(ncas, self.n_alpha)
        ub, rb = string_tables(ncas, self.n_beta)
        self.na, self.nb = len(ua), len(ub)
        self.Dc = self.na * self.nb
        # uint32 strings are stored in int32 tensors (same bits; torch has no uint32 arithmetic)
        self.unrank_a = torch.as_tensor(ua.astype(np.int32)).to(device)
        self.unrank_b = torch.as_tensor(ub.astype(np.int32)).to(device)
        self.rank_a = torch.as_tensor(ra).to(device)
        self.rank_b = torch.as_tensor(rb).to(device)
        self.gates_dev, self.n_gates, self.n_theta = gates_dev, n_gates, n_theta
        self.init_index = init_index
        self._work = {}

    def fits(self):
        """The sector vector (+ gate table) must fit one workgroup's LDS twice (adjoint sweep)."""
        words = self.na + self.nb + 2 * (1 << self.ncas) + self.Dc
        return ((2 * self.Dc + 2 * self.n_gates + 16 + self.n_theta) * 8 + 40 * self.n_gates
                + 4 * words) <= 160 * 1024

    def _tabs(self):
        i32 = torch.int32
        return (dptr(self.unrank_a, i32), dptr(self.unrank_b, i32), dptr(self.rank_a, i32),
                dptr(self.rank_b, i32), self.na, self.nb)

    def work(self, batch):
        if batch not in self._work:
            n = self.lib.oovqe_sector_work_size(self.ncas, self.na, self.nb, batch)
            self._work[batch] = torch.empty(n, dtype=F64, device=self.device)
        return self._work[batch]

    def state(self, theta, dense=False):
        """theta [batch, n_theta] -> psi_c [batch, Dc] (and dense [batch, 2^n] if requested)."""
        batch = theta.shape[0]
        psi_c = torch.empty((batch, self.Dc), dtype=F64, device=self.device)
        psi = (torch.empty((batch, 1 << (2 * self.ncas)), dtype=F64, device=self.device)
               if dense else None)
        check(self.lib.oovqe_sector_state(dptr(theta), self.n_theta, dptr(self.gates_dev, torch.uint8),
                                          self.n_gates, self.ncas, ctypes.c_uint32(self.init_index),
                                          *self._tabs(), batch, dptr(psi_c), dptr(psi), stream_ptr()),
              "oovqe_sector_state")
        return (psi_c, psi) if dense else psi_c

    def rdms(self, psi_c):
        batch = psi_c.shape[0]
        a = self.ncas
        gamma = torch.empty((batch, a, a), dtype=F64, device=self.device)
        Gamma = torch.empty((batch, a, a, a, a), dtype=F64, device=self.device)
        check(self.lib.oovqe_sector_rdms(dptr(psi_c), a, *self._tabs(), batch, dptr(gamma),
                                         dptr(Gamma), dptr(self.work(batch)), stream_ptr()),
              "oovqe_sector_rdms")
        return gamma, Gamma

    def adjoint(self, theta, psi_c, c1, c2):
        """d/dtheta (c1.gamma + c2.Gamma) for each batch element; call after rdms(psi_c)."""
        batch = theta.shape[0]
        dth = torch.empty((batch, self.n_theta), dtype=F64, device=self.device)
        check(self.lib.oovqe_sector_adjoint(dptr(theta), self.n_theta,
                                            dptr(self.gates_dev, torch.uint8), self.n_gates,
                                            self.ncas, *self._tabs(), batch, dptr(psi_c),
                                            dptr(c1.contiguous()), dptr(c2.contiguous()),
                                            dptr(self.work(batch)), dptr(dth), stream_ptr()),
              "oovqe_sector_adjoint")
        return dth
