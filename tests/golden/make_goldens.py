#!/usr/bin/env python3
"""Transcribe the reference's literal known-answer vectors into JSON fixtures.

Run in the build container only (reads /root/reference as TEXT; nothing from the
reference is imported, compiled or executed).  The literal arrays inside the
``pytest.mark.parametrize`` tables of

  * test/test_pqc.py::test_state        (theta -> statevector)        lines 33-263
  * test/test_pqc.py::test_rdms         (theta -> one_rdm, two_rdm)   lines 273-614
  * test/test_oo_energy.py::test_vector_to_skew_symmetric             lines 188-209
  * test/test_oo_energy.py::test_non_redundant_indices                lines 216-227

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


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    states, rdms = extract_pqc(ref_root)
    skew, nonred = extract_oo_energy(ref_root)
    out = {
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
