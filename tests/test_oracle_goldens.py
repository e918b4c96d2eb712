"""Pin the CPU oracle against the reference's own literal known-answer vectors
(test/test_pqc.py::test_state, ::test_rdms; test/test_oo_energy.py:188-231), transcribed into
tests/golden/*.json by tests/golden/make_goldens.py.  Tolerance = the reference's own
``allclose(rtol=1e-5, atol=1e-8)``, plus 1e-8 absolute where the literal carries >= 9 digits."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


def _pqc(c):
    return R.OraclePQC(c["ncas"], c["nelecas"], ansatz=c["ansatz"],
                       n_layers=c["n_layers"] or 3, add_singles=bool(c["add_singles"]))


@pytest.mark.parametrize("case", _load("pqc_states.json"), ids=lambda c: c["source"])
def test_state_golden(case):
    psi = _pqc(case).qnode(torch.tensor(case["theta"], dtype=torch.float64)).numpy()
    ref = np.array(case["state_real"]) + 1j * np.array(case["state_imag"])
    assert np.allclose(psi, ref, rtol=1e-5, atol=1e-8)
    if case["source"].endswith(":36"):      # 5-digit literal
        assert np.abs(psi - ref).max() < 5e-5
    else:
        assert np.abs(psi - ref).max() < 1e-8
    # the simulated state is real up to rounding dust (pqc.py:213-217 relies on it)
    assert np.abs(psi.imag).max() < 1e-14


@pytest.mark.parametrize("case", _load("pqc_rdms.json"), ids=lambda c: c["source"])
def test_rdms_golden(case):
    g1, g2 = _pqc(case).get_rdms(torch.tensor(case["theta"], dtype=torch.float64))
    assert np.allclose(g1.numpy(), np.array(case["one_rdm"]), rtol=1e-5, atol=1e-8)
    assert np.allclose(g2.numpy(), np.array(case["two_rdm"]), rtol=1e-5, atol=1e-8)
    assert np.abs(g1.numpy() - np.array(case["one_rdm"])).max() < 1e-8
    assert np.abs(g2.numpy() - np.array(case["two_rdm"])).max() < 1e-8
    ne = case["nelecas"]
    assert abs(np.trace(g1.numpy()) - ne) < 1e-12
    assert abs(np.einsum("pprr", g2.numpy()) - ne * (ne - 1)) < 1e-12


@pytest.mark.parametrize("case", _load("skew_pack.json"), ids=lambda c: c["source"])
def test_skew_pack_golden(case):
    v = torch.tensor(case["vector"], dtype=torch.float64)
    m = R.vector_to_skew_symmetric(v)
    assert np.array_equal(m.numpy(), np.array(case["matrix"]))
    assert np.array_equal(R.skew_symmetric_to_vector(m).numpy(), np.array(case["vector"]))


@pytest.mark.parametrize("case", _load("nonredundant_idx.json"), ids=lambda c: c["source"])
def test_non_redundant_golden(case):
    idx = R.non_redundant_indices(case["occ_idx"], case["act_idx"], case["virt_idx"],
                                  case["freeze_active"])
    assert np.array_equal(idx, np.array(case["idx_ref"]))


def test_excitation_lists():
    """CAS(4e,3o) doubles / singles as produced by qml.qchem.excitations."""
    s, d = R.excitations(4, 6)
    assert d == [[0, 1, 4, 5], [0, 3, 4, 5], [1, 2, 4, 5], [2, 3, 4, 5]]
    assert s == [[0, 4], [1, 5], [2, 4], [3, 5]]
    assert list(R.hf_state(4, 6)) == [1, 1, 1, 1, 0, 0]
    assert len(R.generalized_pair_doubles(range(16))) == 56


@pytest.mark.parametrize("which", ["small_13_3", "small_13_2_frozen", "small_20_5", "hessian_n43_rng9", "config3_n43"])
def test_oracle_cache_is_what_the_oracle_computes(which, monkeypatch):
    """tests/golden/oracle_cache/*.npz (the oracle's side of the comparisons at the configs[3] shape, minutes of CPU on
    a GPU box) against the oracle itself, recomputed here: a fixture that is stale -- the oracle, the synthetic
    generator or a seed changed -- cannot pass.  (Deterministic on one machine; across machines the autograd
    Hessians agree to rounding: 1e-10.)"""
    import os
    from tests import _oracle_cache as C
    from tests import test_newton_gpu as T
    fn = {"small_13_3": lambda: T.oracle_small_hessians(13, 3, False),
          "small_13_2_frozen": lambda: T.oracle_small_hessians(13, 2, True),
          "small_20_5": lambda: T.oracle_small_hessians(20, 5, False),
          "hessian_n43_rng9": T.oracle_hessian_n43_rng9, "config3_n43": T.oracle_config3_n43}[which]
    stored = fn()
    files_before = sorted(os.listdir(C.CACHE))
    monkeypatch.setattr(C.os.path, "exists", lambda p: False if p.startswith(C.CACHE) else os.path.lexists(p))
    fresh = fn()                                   # (no file seen: the oracle runs; nothing is written)
    assert sorted(os.listdir(C.CACHE)) == files_before
    assert sorted(stored) == sorted(fresh)
    for key in stored:
        a, b = np.asarray(stored[key], dtype=float), np.asarray(fresh[key], dtype=float)
        assert a.shape == b.shape, key
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), key
