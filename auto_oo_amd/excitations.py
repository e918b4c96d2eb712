"""Host-side integer logic of the circuit path: excitation lists and the gate table consumed by
``oovqe_circuit_state`` (include/oovqe.h, ``oovqe_gate_t``).

Reference call sites: ``qml.qchem.excitations`` / ``excitations_to_wires`` / ``hf_state``
(src/auto_oo/pqc.py:123-132), ``generalized_pair_doubles`` (src/auto_oo/ansatze/kUpCCD.py:16-33),
``UCCD.compute_decomposition`` (ansatze/uccd.py:105-114), ``kUpCCD.compute_decomposition``
(ansatze/kUpCCD.py:118-130), ``qml.UCCSD`` (pqc.py:71-73).

Closed form of one FermionicDoubleExcitation(theta; wires1=[s..r], wires2=[q..p]) (PennyLane's
8-layer CNOT-ladder decomposition multiplied out): for every basis state x with b_s=b_r=1,
b_q=b_p=0 and its partner y (those four bits flipped),
    psi'[x] = cos(theta/2) psi[x] + pi sin(theta/2) psi[y]
    psi'[y] = cos(theta/2) psi[y] - pi sin(theta/2) psi[x]
with pi = (-1)^(number of set bits strictly inside wires1 plus strictly inside wires2).
FermionicSingleExcitation(theta; wires=[r..p]) is the same 2x2 rotation on (b_r=1,b_p=0) <->
(b_r=0,b_p=1) with the opposite sign of theta and pi over the wires strictly between.
"""
import ctypes

import numpy as np

from ._lib import GateT


def excitations(electrons, orbitals, delta_sz=0):
    """qml.qchem.excitations with interleaved spins (even wire = alpha)."""
    sz = [0.5 if (i % 2 == 0) else -0.5 for i in range(orbitals)]
    singles = [[r, p] for r in range(electrons) for p in range(electrons, orbitals)
               if sz[p] - sz[r] == delta_sz]
    doubles = [[s, r, q, p]
               for s in range(electrons - 1) for r in range(s + 1, electrons)
               for q in range(electrons, orbitals - 1) for p in range(q + 1, orbitals)
               if (sz[p] + sz[q] - sz[r] - sz[s]) == delta_sz]
    return singles, doubles


def excitations_to_wires(singles, doubles):
    """qml.qchem.excitations_to_wires"""
    s_wires = [list(range(r, p + 1)) for r, p in singles]
    d_wires = [[list(range(s, r + 1)), list(range(q, p + 1))] for s, r, q, p in doubles]
    return s_wires, d_wires


def hf_state(electrons, orbitals):
    """qml.qchem.hf_state"""
    return np.array([1 if i < electrons else 0 for i in range(orbitals)], dtype=int)


def basis_index(occupation):
    """Index of a computational basis state (wire 0 = most significant bit)."""
    return int("".join(str(int(b)) for b in occupation), 2)


def generalized_pair_doubles(wires):
    """ansatze/kUpCCD.py:16-33 (same list as ansatze/uccd.py:117-134)."""
    wires = list(wires)
    return [[wires[r:r + 2], wires[p:p + 2]]
            for r in range(0, len(wires) - 1, 2)
            for p in range(0, len(wires) - 1, 2) if p != r]


def _bit(w, n):
    return 1 << (n - 1 - w)


def _gate(mask_hi, mask_lo, mask_par, theta_idx, sign):
    fm = mask_hi | mask_lo
    pos = [i for i in range(32) if (fm >> i) & 1]
    g = GateT()
    g.mask_hi, g.mask_lo, g.mask_par = mask_hi, mask_lo, mask_par
    g.theta_idx, g.sign, g.nfix = theta_idx, sign, len(pos)
    for i in range(4):
        g.pos[i] = pos[i] if i < len(pos) else 0
    return g


def fde_gate(wires1, wires2, n, theta_idx):
    """FermionicDoubleExcitation(theta[theta_idx], wires1, wires2) on n qubits."""
    wires1, wires2 = list(wires1), list(wires2)
    s, r, q, p = wires1[0], wires1[-1], wires2[0], wires2[-1]
    if len({s, r, q, p}) != 4:
        raise ValueError(f"double excitation needs four distinct end wires, got {wires1}, {wires2}")
    par = 0
    for w in wires1[1:-1] + wires2[1:-1]:
        par |= _bit(w, n)
    return _gate(_bit(s, n) | _bit(r, n), _bit(q, n) | _bit(p, n), par, theta_idx, +1)


def fse_gate(wires, n, theta_idx):
    """FermionicSingleExcitation(theta[theta_idx], wires=[r..p]) on n qubits."""
    wires = list(wires)
    r, p = wires[0], wires[-1]
    par = 0
    for w in wires[1:-1]:
        par |= _bit(w, n)
    return _gate(_bit(r, n), _bit(p, n), par, theta_idx, -1)


def uccd_gates(ncas, nelecas, add_singles=False):
    """Gate list of uccd_circuit (pqc.py:69-76).  UCCD: doubles in order, theta[i].  UCCSD
    (qml.UCCSD): doubles first with theta[len(s_wires)+i], then singles with theta[j]."""
    n = 2 * ncas
    singles, doubles = excitations(nelecas, n)
    s_wires, d_wires = excitations_to_wires(singles, doubles)
    gates = []
    if add_singles:
        for i, (w1, w2) in enumerate(d_wires):
            gates.append(fde_gate(w1, w2, n, len(s_wires) + i))
        for j, w in enumerate(s_wires):
            gates.append(fse_gate(w, n, j))
        n_theta = len(s_wires) + len(d_wires)
    else:
        if not d_wires:
            raise ValueError(f"d_wires lists can not be empty; got pphh={d_wires}")
        for i, (w1, w2) in enumerate(d_wires):
            gates.append(fde_gate(w1, w2, n, i))
        n_theta = len(d_wires)
    return gates, n_theta


def kupccd_gates(ncas, k=1):
    """Gate list of kUpCCD.compute_decomposition (ansatze/kUpCCD.py:118-130): k layers over all
    ordered pairs; weights[layer][i] -> flat index layer * n_pairs + i."""
    n = 2 * ncas
    if n < 4:
        raise ValueError(f"Requires at least four wires; got {n} wires.")
    if k < 1:
        raise ValueError(f"Requires k to be at least 1; got {k}.")
    d_wires = generalized_pair_doubles(range(n))
    gates = []
    for layer in range(k):
        for i, (w1, w2) in enumerate(d_wires):
            gates.append(fde_gate(w1, w2, n, layer * len(d_wires) + i))
    return gates, k * len(d_wires)


def double_excitation_gate(wires, n, theta_idx):
    """qml.DoubleExcitation(phi) on four wires: |0011> -> c|0011> + s|1100>,
    |1100> -> c|1100> - s|0011>  (c, s = cos, sin of phi/2); no fermionic sign."""
    w0, w1, w2, w3 = wires
    return _gate(_bit(w0, n) | _bit(w1, n), _bit(w2, n) | _bit(w3, n), 0, theta_idx, +1)


def orbital_rotation_gates(wires, n, theta_idx):
    """qml.OrbitalRotation(phi) on (q0,q1,q2,q3) = fSWAP[q1,q2] SingleExcitation(phi)[q0,q1]
    SingleExcitation(phi)[q2,q3] fSWAP[q1,q2].  Conjugating an adjacent-qubit Givens rotation by the
    fermionic swap turns it into a Givens rotation between q0 and q2 (resp. q1 and q3) that picks
    up the Jordan-Wigner sign of the qubit in between, so the whole gate is two commuting entries
    of the ordinary gate table, both driven by the same parameter."""
    q0, q1, q2, q3 = wires
    return [_gate(_bit(q0, n), _bit(q2, n), _bit(q1, n), theta_idx, +1),
            _gate(_bit(q1, n), _bit(q3, n), _bit(q2, n), theta_idx, +1)]


def gatefabric_redundant_idx(ncas, nelecas):
    """src/auto_oo/pqc.py:147-153: leading GateFabric parameters that only rotate all-occupied or
    all-virtual orbital pairs of the HF state."""
    n_qubits = 2 * ncas
    if n_qubits > 4:
        red = list(range(0, 2 * (nelecas // 4)))
        if ncas % 2 == 0:
            red += list(range(2 * ((n_qubits - nelecas) // 4), 2 * (n_qubits // 4)))
    else:
        red = []
    return red


def gatefabric_gates(ncas, nelecas, n_layers):
    """Gate list of gatefabric_circuit (pqc.py:79-83,136-160,174-186; qml.GateFabric with
    include_pi=False): per layer the 4-wire blocks [0..3],[4..7],... then [2..5],[6..9],...;
    per block DoubleExcitation(weights[l,i,0]) then OrbitalRotation(weights[l,i,1]).  The
    redundant leading parameters are fixed to zero (theta_idx = -1: the gate is skipped)."""
    n = 2 * ncas
    n_blocks = n // 2 - 1
    full = n_layers * n_blocks * 2
    red = set(gatefabric_redundant_idx(ncas, nelecas))
    params = [x for x in range(full) if x not in red]
    to_user = {f: u for u, f in enumerate(params)}
    wires = list(range(n))
    blocks = [wires[i:i + 4] for i in range(0, n, 4) if i + 4 <= n]
    blocks += [wires[i:i + 4] for i in range(2, n, 4) if i + 4 <= n]
    assert len(blocks) == n_blocks
    gates = []
    for layer in range(n_layers):
        for i, bw in enumerate(blocks):
            f0 = (layer * n_blocks + i) * 2
            gates.append(double_excitation_gate(bw, n, to_user.get(f0, -1)))
            gates.extend(orbital_rotation_gates(bw, n, to_user.get(f0 + 1, -1)))
    return gates, len(params)


def pack_gates(gates):
    """ctypes array (contiguous bytes) of oovqe_gate_t."""
    arr = (GateT * len(gates))()
    for i, g in enumerate(gates):
        arr[i] = g
    return arr


def gates_to_numpy(gates):
    """uint8 view of the packed table (to be moved to the device as a byte tensor)."""
    arr = pack_gates(gates)
    return np.frombuffer(bytes(arr), dtype=np.uint8).copy()


def tril_tables(nao, params_idx):
    """(row, col) int32 tables of the non-redundant kappa entries: position t of the kappa vector
    is element params_idx[t] of the strict lower triangle in np.tril_indices order
    (oo_energy.py:82-86,213-224)."""
    rows, cols = np.tril_indices(nao, -1)
    params_idx = np.asarray(params_idx, dtype=np.int64)
    return rows[params_idx].astype(np.int32), cols[params_idx].astype(np.int32)


def non_redundant_indices(occ_idx, act_idx, virt_idx, freeze_active):
    """oo_energy.py:97-118: positions (in the tril vector) of occ-act, act-virt, occ-virt and,
    unless frozen, act-act rotations."""
    no, na, nv = len(occ_idx), len(act_idx), len(virt_idx)
    nao = no + na + nv
    label = np.zeros(nao, dtype=np.int8)          # 0 occ, 1 act, 2 virt, -1 none
    label[:] = -1
    label[np.asarray(occ_idx, dtype=int)] = 0
    label[np.asarray(act_idx, dtype=int)] = 1
    label[np.asarray(virt_idx, dtype=int)] = 2
    rows, cols = np.tril_indices(nao, -1)
    lr, lc = label[rows], label[cols]
    redundant = ((lr == 1) & (lc == 1) & bool(freeze_active)) | ((lr == 0) & (lc == 0)) \
        | ((lr == 2) & (lc == 2))
    params_idx = np.nonzero(~redundant)[0]
    n_kappa = no * na + na * nv + no * nv + (0 if freeze_active else na * (na - 1) // 2)
    assert n_kappa == len(params_idx)
    return params_idx.astype(int)
