#!/usr/bin/env python3
"""Transcribe the reference's literal known-answer vectors into JSON fixtures.

Run in the build container only (reads /root/reference as TEXT; nothing from the
reference is imported, compiled or executed).  The literal arrays inside the
``pytest.mark.parametrize`` tables of

  * test/test_pqc.py::test_state        (theta -> statevector)        lines 33-263
  * test/test_pqc.py::test_rdms         (theta -> one_rdm, two_rdm)   lines 273-614
  * test/test_oo_energy.py::test_vector_to_skew_symmetric             lines 188-209
  * test/test_oo_energy.py::test_non_redundant_indices                lines 216-227
  * the real-molecule literals (formaldimine / STO-3G; their AO integrals come from
    auto_oo_amd/gaussian.py at test time, the literals below are what they are checked against):
    test/test_moldata_pyscf.py::test_ao_to_oao (S^-1/2, lines 17-85),
    test/test_oo_energy.py::test_mo_ao_to_oao (S^1/2 C_HF, 27-95), ::test_energy_from_mo_coeff
    (orbitals + RDMs -> -92.74923236954386, 241-298), ::test_orbital_optimization (-> the RHF
    energy -92.66372193556138, 318-396), ::test_analytical_derivatives (STO-3G case, 416-473),
    test/test_oo_pqc.py::test_full_derivatives (np_fabric orbitals + theta, 38-84)
  * the printed outputs of the two tutorial notebooks' recorded runs (examples/*.ipynb read as
    JSON): OO-VQE Newton trajectories, Berry-phase-loop energies and state overlaps

are walked with ``ast`` (numbers, lists, unary minus and real+imag sums only) and
written to ``tests/golden/*.json``.  The fixtures are data: inputs and expected outputs.

Usage: python tests/golden/make_goldens.py [/root/reference]
"""
import ast
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _num(node):
    """Literal number / nested list -> python object.  Complex numbers become [re, im]."""
    if isinstance(node, ast.Constant):
        v = node.value
        if isinstance(v, complex):
            return ("c", v.real, v.imag)
        return v
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _num(node.operand)
        sgn = -1.0 if isinstance(node.op, ast.USub) else 1.0
        if isinstance(v, tuple):
            return ("c", sgn * v[1], sgn * v[2])
        return sgn * v if not isinstance(v, int) else int(sgn) * v
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub)):
        a, b = _num(node.left), _num(node.right)
        sgn = 1.0 if isinstance(node.op, ast.Add) else -1.0
        ar, ai = (a[1], a[2]) if isinstance(a, tuple) else (float(a), 0.0)
        br, bi = (b[1], b[2]) if isinstance(b, tuple) else (float(b), 0.0)
        return ("c", ar + sgn * br, ai + sgn * bi)
    if isinstance(node, (ast.List, ast.Tuple)):
        return [_num(e) for e in node.elts]
    if isinstance(node, ast.Call):
        # math.array([...]) / np.array([...]) / math.array([...], like='torch')
        fn = node.func
        name = fn.attr if isinstance(fn, ast.Attribute) else getattr(fn, "id", "")
        if name == "array":
            return _num(node.args[0])
    raise ValueError(f"unsupported literal node: {ast.dump(node)[:80]}")


def _strip_complex(obj):
    """Split nested lists holding ('c', re, im) into (real_list, imag_list)."""
    if isinstance(obj, tuple):
        return obj[1], obj[2]
    if isinstance(obj, list):
        parts = [_strip_complex(o) for o in obj]
        return [p[0] for p in parts], [p[1] for p in parts]
    return float(obj), 0.0


def _param_tables(path):
    """Yield (test_function_name, argnames, list-of-case-nodes) for each parametrize table."""
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read())
    for node in tree.body:
        if not isinstance(node, ast.FunctionDef):
            continue
        for dec in node.decorator_list:
            if not (isinstance(dec, ast.Call) and isinstance(dec.func, ast.Attribute)
                    and dec.func.attr == "parametrize"):
                continue
            names = dec.args[0]
            if isinstance(names, ast.Tuple):
                argnames = [e.value for e in names.elts]
            else:
                argnames = [s.strip() for s in names.value.split(",")]
            yield node.name, argnames, dec.args[1].elts


def _plain(node):
    if isinstance(node, ast.Constant):
        return node.value
    return _num(node)


def extract_pqc(ref_root):
    path = os.path.join(ref_root, "test", "test_pqc.py")
    states, rdms = [], []
    for fname, argnames, cases in _param_tables(path):
        for case in cases:
            vals = dict(zip(argnames, case.elts))
            rec = {
                "ncas": _plain(vals["ncas"]),
                "nelecas": _plain(vals["nelecas"]),
                "add_singles": _plain(vals["add_singles"]),
                "ansatz": _plain(vals["ansatz"]),
                "n_layers": _plain(vals["n_layers"]),
                "theta": _strip_complex(_num(vals["theta"]))[0],
                "source": f"test/test_pqc.py:{case.lineno}",
            }
            if fname == "test_state":
                re, im = _strip_complex(_num(vals["state_ref"]))
                rec["state_real"], rec["state_imag"] = re, im
                states.append(rec)
            elif fname == "test_rdms":
                rec["one_rdm"] = _strip_complex(_num(vals["one_rdm_ref"]))[0]
                rec["two_rdm"] = _strip_complex(_num(vals["two_rdm_ref"]))[0]
                rdms.append(rec)
    return states, rdms


def extract_oo_energy(ref_root):
    path = os.path.join(ref_root, "test", "test_oo_energy.py")
    skew, nonred = [], []
    for fname, argnames, cases in _param_tables(path):
        if fname == "test_vector_to_skew_symmetric":
            for case in cases:
                vals = dict(zip(argnames, case.elts))
                skew.append({
                    "vector": _strip_complex(_num(vals["vector"]))[0],
                    "matrix": _strip_complex(_num(vals["matrix_ref"]))[0],
                    "source": f"test/test_oo_energy.py:{case.lineno}",
                })
        elif fname == "test_non_redundant_indices":
            for case in cases:
                vals = dict(zip(argnames, case.elts))
                nonred.append({
                    "occ_idx": _num(vals["occ_idx"]),
                    "act_idx": _num(vals["act_idx"]),
                    "virt_idx": _num(vals["virt_idx"]),
                    "freeze_active": _plain(vals["freeze_active"]),
                    "idx_ref": [int(v) for v in _strip_complex(_num(vals["idx_ref"]))[0]],
                    "source": f"test/test_oo_energy.py:{case.lineno}",
                })
    return skew, nonred


def _geometry(node):
    """get_formal_geo(alpha, phi) call -> {"formal_geo": [alpha, phi]}; a string literal stays."""
    if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "get_formal_geo":
        return {"formal_geo": [_plain(a) for a in node.args]}
    return {"atoms": _plain(node)}


def extract_molecule_cases(ref_root):
    """Every STO-3G case whose expected values are literals in the reference's tests."""
    wanted = {
        ("test_moldata_pyscf.py", "test_ao_to_oao"): ["oao_coeff_ref"],
        ("test_oo_energy.py", "test_mo_ao_to_oao"): ["hf_oao_coeff_ref"],
        ("test_oo_energy.py", "test_energy_from_mo_coeff"): ["mo_coeff", "one_rdm", "two_rdm", "e_ref"],
        ("test_oo_energy.py", "test_orbital_optimization"): ["mo_coeff", "one_rdm", "two_rdm", "e_ref"],
        ("test_oo_energy.py", "test_analytical_derivatives"): ["mo_coeff", "one_rdm", "two_rdm"],
        ("test_oo_pqc.py", "test_full_derivatives"): ["oao_mo_coeff", "theta"],
    }
    out = []
    for (fname, test), arrays in wanted.items():
        path = os.path.join(ref_root, "test", fname)
        for tname, argnames, cases in _param_tables(path):
            if tname != test:
                continue
            for case in cases:
                vals = dict(zip(argnames, case.elts))
                if _plain(vals["basis"]) != "sto-3g":
                    continue                        # cc-pVDZ needs d shells: no integrals here
                rec = {"test": test, "source": f"test/{fname}:{case.lineno}",
                       "geometry": _geometry(vals["geometry"]), "basis": "sto-3g"}
                for key in ("ncas", "nelecas", "n_layers", "freeze_active", "check_hess"):
                    if key in vals:
                        rec[key] = _plain(vals[key])
                for key in arrays:
                    rec[key] = _strip_complex(_num(vals[key]))[0]
                out.append(rec)
    return out


def _cell_outputs(nb_path):
    """(source text, printed text) of every code cell of a notebook (read as JSON data)."""
    with open(nb_path, "r", encoding="utf-8") as fh:
        nb = json.load(fh)
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        text = "".join("".join(o.get("text", [])) for o in cell.get("outputs", []) if "text" in o)
        yield "".join(cell["source"]), text


def _floats_after(text, prefix_re):
    import re
    return [float(m) for m in re.findall(prefix_re + r"\s*(-?\d+\.\d+(?:[eE][-+]?\d+)?)", text)]


def extract_notebook_runs(ref_root):
    """Printed results of the two tutorial notebooks (the reference's own recorded runs on
    formaldimine / STO-3G): the OO-VQE Newton trajectory of examples/Tutorial_auto_oo.ipynb and the
    whole Berry-phase loop of examples/Tutorial_Berry_phase.ipynb (pre-optimisation, one damped
    Newton step per loop point, state overlaps under the active-space orbital rotation).  The run
    settings are the constants assigned in the notebooks' cells, transcribed next to the numbers."""
    out = {}
    # ---- Tutorial_auto_oo: CAS(4e,3o), np_fabric 2 layers, (alpha, phi) = (140, 80)
    for src, text in _cell_outputs(os.path.join(ref_root, "examples", "Tutorial_auto_oo.ipynb")):
        if "mol.casscf.e_tot" in src and "Hartree-Fock energy" in src and "CASCI energy" in text:
            out.setdefault("tutorial_auto_oo", {})["printed_hf_casci_casscf"] = [
                _floats_after(text, "Hartree-Fock energy:")[0], _floats_after(text, "CASCI energy:")[0],
                _floats_after(text, "CASSCF energy:")[0]]
        if "full_optimization(theta_zero)" in src:
            rec = out.setdefault("tutorial_auto_oo", {})
            rec.update({"source": "examples/Tutorial_auto_oo.ipynb: oo_pqc.full_optimization(theta_zero)",
                        "formal_geo": [140, 80], "basis": "sto-3g", "ncas": 3, "nelecas": 4,
                        "ansatz": "np_fabric", "n_layers": 2, "freeze_active": True,
                        "energies": _floats_after(text, r"iter = \d+, energy ="),
                        "E_fin": _floats_after(text, "E_fin =")[0]})
    # ---- Tutorial_Berry_phase: CAS(2e,2o), np_fabric 1 layer, 10 points on a loop
    rec = {"source": "examples/Tutorial_Berry_phase.ipynb", "basis": "sto-3g", "ncas": 2, "nelecas": 2,
           "ansatz": "np_fabric", "n_layers": 1, "freeze_active": True,
           "origin": [130, 89.9], "radius": [10, 10], "t0": 0.0, "phase_pi_over": 20, "n_points": 10}
    for src, text in _cell_outputs(os.path.join(ref_root, "examples", "Tutorial_Berry_phase.ipynb")):
        if "full_optimization(theta0)" in src:
            rec["preopt_energies"] = _floats_after(text, r"iter = \d+, energy =")
            rec["preopt_E_fin"] = _floats_after(text, "E_fin =")[0]
            rec["preopt_lowest_hessian_eigenvalue"] = _floats_after(text, "lowest Hessian eigenvalue:")[0]
            rec["preopt_casscf_energy"] = _floats_after(text, "Casscf energy =")[0]
        if "damped_newton_step" in src and "Energy at step" in text:
            rec["loop_energies"] = _floats_after(text, r"Energy at step \d+:")
        if "overlaps.append" in src:
            rec["overlaps"] = _floats_after(text, r"Overlap \S+ \| G \| \d+\S:")
            rec["final_overlap"] = _floats_after(text, r"Final overlap \S+ \| G \| \d+\S:")[0]
    out["tutorial_berry_phase"] = rec
    return out


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    states, rdms = extract_pqc(ref_root)
    skew, nonred = extract_oo_energy(ref_root)
    out = {
        "molecule_cases.json": extract_molecule_cases(ref_root),
        "notebook_runs.json": extract_notebook_runs(ref_root),
        "pqc_states.json": states,
        "pqc_rdms.json": rdms,
        "skew_pack.json": skew,
        "nonredundant_idx.json": nonred,
    }
    for name, data in out.items():
        with open(os.path.join(HERE, name), "w", encoding="utf-8") as fh:
            json.dump(data, fh, indent=1)
        print(f"{name}: {len(data)} cases")


if __name__ == "__main__":
    main()
