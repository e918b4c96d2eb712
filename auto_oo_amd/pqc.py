"""Parameterized_circuit: statevector ansatz + RDMs on the MI355X.

Drop-in for the reference's ``auto_oo.Parameterized_circuit`` (src/auto_oo/pqc.py:86-235) for
``ansatz='ucc'`` (UCCD, and UCCSD with ``add_singles=True``), ``ansatz='np_fabric'`` (GateFabric)
plus ``ansatz='kupccd'`` (the reference defines the kUpCCD operator, ansatze/kUpCCD.py:36-154, but
never wires it into ``Parameterized_circuit``).  Where the reference accepts a custom PennyLane
QNode as ``ansatz`` (pqc.py:162-163) this class accepts a custom GATE TABLE: a list of gates built
with ``auto_oo_amd.excitations`` (``fde_gate``, ``fse_gate``, ``double_excitation_gate``,
``orbital_rotation_gates``), or a callable ``(ncas, nelecas) -> gates``; there is no PennyLane
here.  The PennyLane device argument ``dev`` is accepted and ignored: the
state is produced by ``oovqe_circuit_state`` and the RDMs by ``oovqe_rdms*`` (include/oovqe.h).
"""
import warnings

import numpy as np
import torch

from . import _lib, excitations as X, ops
from .autodiff import VectorWithJacobian, needs_autodiff
from .sector import SectorEngine


class Parameterized_circuit():
    """Parameterized quantum circuit class.  Defined by an active space of nelecas electrons in
    ncas orbitals.  ``qnode(theta)`` outputs the quantum state, ``get_rdms`` the one- and two-RDMs."""

    def __init__(self, ncas, nelecas, dev=None, ansatz='ucc', n_layers=3,
                 add_singles=False, interface='torch', diff_method='backprop', k=1):
        """
        Args (pqc.py:91-111):
            ncas: Number of active orbitals
            nelecas: Number of active electrons
            dev: ignored (kept for signature compatibility with the PennyLane-based reference)
            ansatz: 'ucc' (UCCD / UCCSD), 'np_fabric' (GateFabric) or 'kupccd'
            n_layers: layers of an 'np_fabric' ansatz
            add_singles: add UCC single excitations to a 'ucc' ansatz
            k: number of kUpCCD layers (``ansatz='kupccd'`` only)
        """
        if interface != 'torch':
            raise ValueError("auto_oo_amd supports interface='torch' only (PyTorch-ROCm tensors)")
        self.ncas = ncas
        self.nelecas = nelecas
        self.n_qubits = 2 * ncas
        self.dev = dev
        self.add_singles = add_singles
        self.interface = interface
        self.device = _lib.require_device()

        self.e_pq = None      # kept for attribute compatibility (pqc.py:118-119); unused:
        self.e_pqrs = None    # the RDM kernels apply E_pq by bit operations
        self.up_then_down = False

        if ansatz == 'ucc':
            self.singles, self.doubles = X.excitations(nelecas, self.n_qubits)
            self.s_wires, self.d_wires = X.excitations_to_wires(self.singles, self.doubles)
            self._gates, self.theta_shape = X.uccd_gates(ncas, nelecas, add_singles)
        elif ansatz == 'kupccd':
            self.k = k
            self.d_wires = X.generalized_pair_doubles(range(self.n_qubits))
            self._gates, n_theta = X.kupccd_gates(ncas, k)
            self.theta_shape = n_theta
        elif ansatz == 'np_fabric':
            # pqc.py:136-160: GateFabric layers; the leading parameters that are redundant when
            # starting from the HF state are fixed to zero
            self.n_layers = n_layers
            self.full_theta_shape = (n_layers, self.n_qubits // 2 - 1, 2)
            self.redundant_idx = X.gatefabric_redundant_idx(ncas, nelecas)
            self._gates, n_theta = X.gatefabric_gates(ncas, nelecas, n_layers)
            self.params_idx = torch.as_tensor(
                np.array([x for x in range(int(np.prod(self.full_theta_shape)))
                          if x not in self.redundant_idx]), device=self.device)
            self.theta_shape = n_theta
        elif callable(ansatz) or isinstance(ansatz, (list, tuple)):
            # custom circuit (the counterpart of a user QNode, pqc.py:162-163): a gate table over the
            # engine's gate set, applied to the Hartree-Fock state in list order; parameter k of
            # theta drives every gate whose theta_idx is k
            gates = ansatz(ncas, nelecas) if callable(ansatz) else ansatz
            if isinstance(gates, tuple) and len(gates) == 2 and isinstance(gates[1], (int, np.integer)):
                gates = gates[0]                      # (gates, n_theta) as the built-in builders return
            gates = list(gates)
            if not gates or not all(isinstance(g, X.GateT) for g in gates):
                raise ValueError("a custom ansatz is a non-empty list of excitations.GateT gates")
            self._gates = gates
            self.theta_shape = 1 + max(int(g.theta_idx) for g in gates)
            if self.theta_shape < 1:
                raise ValueError("a custom ansatz needs at least one parameterised gate")
            ansatz = 'custom'
        else:
            raise ValueError(f"unknown ansatz {ansatz!r}: expected 'ucc', 'np_fabric', 'kupccd' or a gate table")
        self.ansatz = ansatz
        self.hfstate = X.hf_state(nelecas, self.n_qubits)
        self.wires = range(self.n_qubits)
        self._init_index = X.basis_index(self.hfstate)
        self._n_gates = len(self._gates)
        self._gates_dev = torch.as_tensor(X.gates_to_numpy(self._gates)).to(self.device)
        # (N_alpha, N_beta)-sector engine: one-launch circuits + reverse-mode gradients for active
        # spaces beyond the small one-workgroup kernel (e.g. kUpCCD CAS(8e,8o), 16 qubits)
        self._sector = SectorEngine(ncas, self.hfstate, self._gates_dev, self._n_gates,
                                    int(np.prod(self.theta_shape)), self._init_index, self.device)
        self._use_sector = self.n_qubits > 10 and self._sector.fits()
        self.qnode = self._qnode

    # ---- internal real-amplitude entry points ---------------------------------------------------
    def _theta2d(self, theta):
        if not (isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dtype == torch.float64
                and theta.is_contiguous()):
            theta = torch.as_tensor(theta)
            if theta.dtype != torch.float64:
                warnings.warn("Input a single precision theta. Only double precision is supported.")
            theta = ops.as_device(theta, self.device)
        theta = theta.detach().reshape(1, -1)
        if theta.shape[1] != int(np.prod(self.theta_shape)):
            raise ValueError(f"Weights tensor must be of shape {(int(np.prod(self.theta_shape)),)}; "
                             f"got {tuple(theta.shape[1:])}.")
        return theta

    def state_real(self, theta, tangents=False):
        """Real amplitudes psi [D] (and tangents d psi/d theta [n_theta, D]) on the device."""
        th = self._theta2d(theta)
        if self._use_sector and not tangents:
            return self._sector.state(th, dense=True)[1][0]
        res = ops.circuit_state(th, self._gates_dev, self._n_gates, self.n_qubits,
                                self._init_index, tangents=tangents)
        if tangents:
            return res[0][0], res[1][0]
        return res[0]

    def rdms_with_derivatives(self, theta):
        """gamma [1+n_theta, a, a], Gamma [1+n_theta, a,a,a,a]: set 0 = RDMs of psi(theta), set k =
        d/dtheta_k (what autograd yields in the reference, oo_pqc.py:86-95,113-119)."""
        th = self._theta2d(theta)
        if self._use_sector:
            return self._sector.rdms_with_derivatives(th, self._gates)
        gamma, Gamma = ops.circuit_rdms(th, self._gates_dev, self._n_gates, self.n_qubits,
                                        self.ncas, self._init_index, tangents=True)
        return gamma[0], Gamma[0]

    def _rdms_and_jacobians(self, theta):
        gamma, Gamma = self.rdms_with_derivatives(theta)
        return (gamma[0], Gamma[0]), (gamma[1:], Gamma[1:])

    # ---- reference API ----------------------------------------------------------------------------
    def _qnode(self, theta):
        """State as a complex128 vector of length 2^n (pqc.py:133,165-172).  The UCC(S)D / kUpCCD
        state is real; the imaginary part is exactly zero here (1e-17 dust in the reference)."""
        return self.state_real(theta).to(torch.complex128)

    def uccd_state(self, theta):
        return self._qnode(theta)

    def gatefabric_state(self, theta):
        """pqc.py:174-186"""
        return self._qnode(theta)

    def init_zeros(self):
        """pqc.py:188-190"""
        return torch.zeros(self.theta_shape, dtype=torch.float64, device=self.device)

    def get_rdms_from_state(self, state, restricted=True):
        """pqc.py:192-218: gamma_pq = (state @ (E_pq @ state)).real, Gamma_pqrs likewise with
        e_pqrs = E_pq E_rs - delta_qr E_ps -- the bilinear form (no conjugation) of the
        reference: Re[psi^T E psi] = Re(psi)^T E Re(psi) - Im(psi)^T E Im(psi).
        ``restricted=False``: the spin-orbital RDMs <a+_p a_q>, <a+_p a+_q a_r a_s> over the 2 ncas
        spin orbitals (utils/active_space.py:44-52,79-82)."""
        state = torch.as_tensor(state)
        if restricted:
            kernel = lambda v: ops.rdms(v, v, self.ncas)                   # noqa: E731
        else:
            kernel = lambda v: ops.spin_rdms(v, v, self.n_qubits)          # noqa: E731
        if state.is_complex():
            re = ops.as_device(state.real, self.device).reshape(1, -1)
            im = ops.as_device(state.imag, self.device).reshape(1, -1)
            g1, g2 = kernel(re)
            if bool((im != 0).any()):
                i1, i2 = kernel(im)
                g1, g2 = g1 - i1, g2 - i2
            return g1[0], g2[0]
        g1, g2 = kernel(ops.as_device(state, self.device).reshape(1, -1))
        return g1[0], g2[0]

    def get_rdms(self, theta, restricted=True):
        """pqc.py:220-221.  Differentiable by torch with respect to theta (first order): the
        Jacobians are the derivative RDMs of the tangent-state kernels."""
        if not restricted:
            return self.get_rdms_from_state(self.state_real(theta), restricted=False)
        if needs_autodiff(theta):
            th = torch.as_tensor(theta).to(device=self.device, dtype=torch.float64).contiguous()
            out = VectorWithJacobian.apply(self._rdms_and_jacobians, th)
            return out[0], out[1]
        th = self._theta2d(theta)
        if self._use_sector:
            g1, g2 = self._sector.rdms(self._sector.state(th))
            return g1[0], g2[0]
        g1, g2 = ops.circuit_rdms(th, self._gates_dev, self._n_gates, self.n_qubits, self.ncas,
                                  self._init_index, tangents=False)
        return g1[0, 0], g2[0, 0]

    def draw_circuit(self, theta):
        """pqc.py:223-225 (text listing of the excitation gates instead of qml.draw)."""
        lines = []
        for g in self._gates:
            lines.append(f"Givens(theta[{g.theta_idx}]*{g.sign}/2) hi={g.mask_hi:0{self.n_qubits}b} "
                         f"lo={g.mask_lo:0{self.n_qubits}b} parity={g.mask_par:0{self.n_qubits}b}")
        return "\n".join(lines)

    def init_e_pq(self, restricted=True):
        """pqc.py:227-230: nothing to build, E_pq is applied by bit operations on the device."""

    def init_e_pqrs(self, restricted=True):
        """pqc.py:232-235: see init_e_pq."""
