"""Active-space reductions of full MO integrals (API helpers).

Drop-ins for the reference's ``utils/active_space.py:111-212`` (``active_space_integrals``,
``molecular_hamiltonian_coefficients``, exported at package level, src/auto_oo/__init__.py:19-28) for
callers that already hold full ``[N,N]`` / ``[N,N,N,N]`` MO tensors.  The evaluation path itself
never forms those tensors: ``OO_energy.get_active_integrals`` / ``oovqe_cas_eval`` produce the same
``(c0, c1, c2)`` straight from the AO integrals.  Tensors stay on the device they come on; the
reductions are diagonal views and index selections (no N^4 temporaries beyond the input).
"""
import torch


def _idx(values, device):
    return torch.as_tensor(list(map(int, values)), dtype=torch.long, device=device)


def active_space_integrals(one_body_integrals, two_body_integrals, occ_idx, act_idx):
    """active_space.py:111-174: core constant, effective one-body integrals and the two-body
    integrals of the active space (chemist order):
        E_core  = 2 sum_i h_ii + sum_ij (2 (ii|jj) - (ij|ji))
        h'_pq   = h_pq + sum_i (2 (pq|ii) - (pi|iq))              p, q active
        g'_pqrs = (pq|rs)                                         p, q, r, s active"""
    h = torch.as_tensor(one_body_integrals)
    g = torch.as_tensor(two_body_integrals).to(h.device)
    occ, act = _idx(occ_idx, h.device), _idx(act_idx, h.device)
    coul = g.diagonal(dim1=2, dim2=3).index_select(2, occ).sum(dim=2)      # sum_i (pq|ii)
    exch = g.diagonal(dim1=1, dim2=2).index_select(2, occ).sum(dim=2)      # sum_i (pi|iq)  [p, q]
    core = (2.0 * h.diagonal().index_select(0, occ).sum()
            + 2.0 * coul.diagonal().index_select(0, occ).sum()
            - exch.diagonal().index_select(0, occ).sum())
    eff = (h + 2.0 * coul - exch).index_select(0, act).index_select(1, act)
    g_act = g.index_select(0, act).index_select(1, act).index_select(2, act).index_select(3, act)
    return core, eff, g_act


def molecular_hamiltonian_coefficients(nuclear_repulsion, one_body_integrals, two_body_integrals,
                                       occ_idx=None, act_idx=None):
    """active_space.py:177-212: (E_constant, c1, c2 = g/2); without index lists the full tensors are
    passed through."""
    if occ_idx is None and act_idx is None:
        return nuclear_repulsion, one_body_integrals, 0.5 * torch.as_tensor(two_body_integrals)
    core, eff, g_act = active_space_integrals(one_body_integrals, two_body_integrals, occ_idx, act_idx)
    return core + nuclear_repulsion, eff, 0.5 * g_act
