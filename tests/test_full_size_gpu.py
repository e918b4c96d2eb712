"""Properties at BASELINE.json's full sizes, where the oracle is too slow to run: the N = 200
four-index transform (configs[2]) and the 64-geometry batched evaluation of the bench workload
(configs[1] shape).  Size-independent properties only: round trips, linearity, invariants,
permutation equivariance, derivative consistency."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _orthogonal(n, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return torch.tensor(q, device=DEV)


def test_transform_n200_round_trip_linearity_invariant():
    from auto_oo_amd import ops
    N = 200
    gen = torch.Generator(device=DEV).manual_seed(7)
    g = torch.rand((N, N, N, N), dtype=torch.float64, device=DEV, generator=gen) - 0.5
    Q = _orthogonal(N, 1)
    QT = Q.T.contiguous()
    out = torch.empty_like(g)
    work = torch.empty_like(g)
    back = torch.empty_like(g)
    ops.general_4index_transform(g, Q, Q, Q, Q, out=out, work=work)
    # invariant of an orthogonal transform: sum_pq g[p,p,q,q] (pair traces) and the Frobenius norm
    tr_in = torch.einsum("ppqq->", g)
    tr_out = torch.einsum("ppqq->", out)
    assert abs((tr_out - tr_in).item()) < 1e-8 * max(1.0, abs(tr_in.item()))
    n_in, n_out = torch.linalg.vector_norm(g), torch.linalg.vector_norm(out)
    assert abs((n_out - n_in).item()) < 1e-11 * n_in.item()
    # round trip: transforming back with Q^T restores the tensor
    ops.general_4index_transform(out, QT, QT, QT, QT, out=back, work=work)
    assert (back - g).abs().max().item() < 1e-11
    # linearity in the tensor argument (independent coefficient matrices per index)
    C = [_orthogonal(N, 10 + i) for i in range(4)]
    g2 = torch.rand((N, N, N, N), dtype=torch.float64, device=DEV, generator=gen) - 0.5
    ops.general_4index_transform(g, *C, out=out, work=work)       # T(g)
    ops.general_4index_transform(g2, *C, out=back, work=work)     # T(g2)
    out.mul_(0.75).add_(back)                                      # 0.75 T(g) + T(g2)
    g.mul_(0.75).add_(g2)                                          # 0.75 g + g2 (in place: memory)
    ops.general_4index_transform(g, *C, out=back, work=work)
    scale = out.abs().max().item()
    assert (back - out).abs().max().item() < 1e-12 * max(1.0, scale)


def test_batched_evaluation_64_geometries_properties():
    """configs[1] shape (N = 43, CAS(4e,3o), UCCD), 64 geometries in one call: permutation
    equivariance (bitwise), energy-only == first column of energy+gradient, and dE/dtheta
    against central differences of the energy for every geometry."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec, G = 43, 3, 4, 16, 64
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    mols, coeffs = [], []
    for g in range(G):
        P = synthetic_problem(N, 9000 + g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec))
        coeffs.append(P["oao_mo_coeff"])
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    rng = np.random.default_rng(5)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)), device=DEV)
    eg = batch.energy_and_gradient(thetas).clone()
    assert torch.isfinite(eg).all()
    assert torch.equal(batch.energy(thetas), eg[:, 0])
    # the same geometries in another order: same numbers, permuted (each geometry is evaluated
    # by its own workgroups with a fixed summation order)
    perm = torch.tensor(rng.permutation(G))
    batch_p = aoo.OO_pqc_batch(pqc, [mols[i] for i in perm], ncas, nelecas,
                               oao_mo_coeffs=[coeffs[i] for i in perm])
    eg_p = batch_p.energy_and_gradient(thetas[perm.to(DEV)].contiguous())
    assert torch.equal(eg_p, eg[perm.to(DEV)])
    # dE/dtheta_k == central difference of E, all geometries at once
    h = 1e-5
    for k in range(pqc.theta_shape):
        tp, tm = thetas.clone(), thetas.clone()
        tp[:, k] += h
        tm[:, k] -= h
        fd = (batch.energy(tp) - batch.energy(tm)) / (2 * h)
        assert (fd - eg[:, 1 + k]).abs().max().item() < 2e-7


def test_batched_evaluation_256_geometries_matches_smaller_batches():
    """The bench's launch shape (N = 43, 256 geometries per call: one workgroup per geometry in the
    N^4 pass, four burst phases, the two-workgroups-per-CU build of the q->x / p->n kernel):
    geometry g of the big batch equals geometry g of a 64-geometry batch holding the same
    molecules, and a single un-batched evaluation."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec, G = 43, 3, 4, 16, 256
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    base = [synthetic_problem(N, 9500 + g) for g in range(6)]
    mols = [aoo.Moldata(base[g % 6]["int1e_ao"], base[g % 6]["int2e_ao"], base[g % 6]["overlap"],
                        base[g % 6]["nuc"] + 0.001 * g, nelec) for g in range(G)]
    coeffs = [base[g % 6]["oao_mo_coeff"] for g in range(G)]
    big = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    assert big.eri_flags == 3 and big._eri_packed is not None
    rng = np.random.default_rng(6)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)), device=DEV)
    eg = big.energy_and_gradient(thetas).clone()
    assert torch.isfinite(eg).all()
    small = aoo.OO_pqc_batch(pqc, mols[100:164], ncas, nelecas, oao_mo_coeffs=coeffs[100:164])
    eg_s = small.energy_and_gradient(thetas[100:164].contiguous())
    assert (eg_s - eg[100:164]).abs().max().item() < 1e-11
    for g in (0, 131, 255):
        single = aoo.OO_pqc(pqc, mols[g], ncas, nelecas, oao_mo_coeff=coeffs[g])
        E, grad = single.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - E.item()) < 1e-11
        assert (eg[g, 1:] - grad).abs().max().item() < 1e-11


def test_deferred_evaluations_equal_the_in_order_calls_bit_for_bit():
    """OO_pqc_batch.evaluate_deferred: independent 256-geometry calls in flight on the library's two side streams
    (the tail of one call under the N^4 sweep of the next), with different thetas per call, against the same calls
    made in order on the current stream -- the same launches per call, so the same bits; with the N^4 sweeps of
    the two streams ordered one after the other (default) and free-running."""
    import auto_oo_amd as aoo
    from auto_oo_amd import _lib
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec, G = 43, 3, 4, 16, 256
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    base = [synthetic_problem(N, 9600 + g) for g in range(4)]
    mols = [aoo.Moldata(base[g % 4]["int1e_ao"], base[g % 4]["int2e_ao"], base[g % 4]["overlap"],
                        base[g % 4]["nuc"] + 0.001 * g, nelec) for g in range(G)]
    big = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=[base[g % 4]["oao_mo_coeff"] for g in range(G)])
    rng = np.random.default_rng(16)
    sets = [torch.tensor(rng.uniform(0, 2 * np.pi, (G, pqc.theta_shape)), device=DEV) for _ in range(5)]
    ref = [big.energy_and_gradient(th).clone() for th in sets]
    assert not torch.equal(ref[0], ref[1])
    with _lib.debug_options(stage1_free_run=1):        # (sweeps of different streams not ordered: the same bits)
        free = [big.energy_and_gradient(th, defer=True) for th in sets]
        inorder = [p.result().clone() for p in free]
    pend = [big.energy_and_gradient(th, defer=True) for th in sets]
    pend += [big.evaluate_deferred(sets[0], derivatives=False, count=100)]
    got = [p.result() for p in pend]
    for r, a, b in zip(ref, inorder, got):
        assert torch.equal(r, a) and torch.equal(r, b)
    assert torch.equal(got[5][:, 1], ref[0][:100, 0])
    # a second round reuses both workspaces while results of the first are still referenced
    again = [big.energy_and_gradient(th, defer=True) for th in reversed(sets)]
    for r, p in zip(reversed(ref), again):
        assert torch.equal(r, p.result())


def test_expm_n200_orthogonality_and_inverse():
    """configs[2]: expm(-K) of a 200 x 200 skew-symmetric K is orthogonal, expm(K) is its inverse,
    and the exponential of a sum of commuting generators (K, 0.5 K) factorises."""
    from auto_oo_amd import ops
    N = 200
    rng = np.random.default_rng(11)
    a = rng.standard_normal((N, N)) * 0.05
    K = torch.tensor(a - a.T, device=DEV)
    U = ops.expm(K, sign=-1.0)
    Ui = ops.expm(K, sign=1.0)
    eye = torch.eye(N, dtype=torch.float64, device=DEV)
    assert (U.T @ U - eye).abs().max().item() < 1e-12
    assert (U @ Ui - eye).abs().max().item() < 1e-12
    U15 = ops.expm((1.5 * K).contiguous(), sign=-1.0)
    Uh = ops.expm((0.5 * K).contiguous(), sign=-1.0)
    assert (U15 - U @ Uh).abs().max().item() < 1e-12


@pytest.mark.parametrize("k", [1, 2])
def test_kupccd_cas88_state_rdm_gradient_properties(k):
    """configs[4]: kUpCCD CAS(8e,8o) (16 qubits, 4900-determinant sector), k = 1 and 2 layers (56 /
    112 thetas, ansatze/kUpCCD.py:16-33): normalisation, RDM sum rules, and the reverse-mode
    theta-gradient against central differences of the energy."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    ncas, nelecas, nelec, N = 8, 8, 16, 43
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=k)
    assert int(pqc.theta_shape) == 56 * k
    rng = np.random.default_rng(2 + k)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, int(pqc.theta_shape)), device=DEV)
    psi = pqc.qnode(th)
    assert abs(torch.linalg.vector_norm(psi).item() - 1.0) < 1e-12
    g1, g2 = pqc.get_rdms(th)
    assert abs(torch.trace(g1).item() - nelecas) < 1e-10
    assert (g1 - g1.T).abs().max().item() < 1e-12
    assert abs(torch.einsum("ppqq->", g2).item() - nelecas * (nelecas - 1)) < 1e-9
    # sum_q Gamma_pqqs-type partial trace: sum_rs delta_rs Gamma[p,q,r,s] = (N - 1) gamma[p,q]
    assert (torch.einsum("pqrr->pq", g2) - (nelecas - 1) * g1).abs().max().item() < 1e-9
    P = synthetic_problem(N, 20265)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    E, grad = oo.energy_and_gradient(th)
    assert abs(E.item() - oo.energy_from_parameters(th).item()) < 1e-10
    h = 1e-5
    order = torch.argsort(grad[:int(pqc.theta_shape)].abs(), descending=True)[:3]
    for j in order.tolist():
        tp, tm = th.clone(), th.clone()
        tp[j] += h
        tm[j] -= h
        fd = (oo.energy_from_parameters(tp).item() - oo.energy_from_parameters(tm).item()) / (2 * h)
        assert abs(fd - grad[j].item()) < 1e-6 * max(1.0, abs(fd))


def test_oo_evaluation_n96_cas66_vs_oracle():
    """Beyond the packed-triangle envelope (N > 48, M = 16): one OO evaluation at N = 96 with a
    CAS(6e,6o) active space (n_occ = 10) on the streaming T2 path against the oracle -- energy,
    CAS coefficients and the analytic orbital gradient for the engine's own RDMs (1e-9 / 1e-8)."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    from oracle import cpu_ref as R
    N, ncas, nelecas, nelec = 96, 6, 6, 26
    P = synthetic_problem(N, 20296)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    assert oo._M == 16 and oo.n_kappa == N * (N - 1) // 2 - 10 * 9 // 2 - 80 * 79 // 2
    theta = torch.tensor(np.random.default_rng(4).uniform(0, 2 * np.pi, pqc.theta_shape))
    E, grad = oo.energy_and_gradient(theta)
    g1, g2 = pqc.get_rdms(theta)
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    ooo = R.OracleOOEnergy(omol, ncas, nelecas, P["oao_mo_coeff"])
    e_ref = ooo.energy_from_mo_coeff(ooo.mo_coeff, g1.cpu(), g2.cpu()).item()
    assert abs(E.item() - e_ref) < 1e-9
    gk_ref = ooo.kappa_matrix_to_vector(ooo.analytic_gradient(g1.cpu(), g2.cpu()))
    assert (grad[pqc.theta_shape:].cpu() - gk_ref).abs().max() < 1e-8
    c0, c1, c2 = oo.get_active_integrals(oo.mo_coeff)
    r0, r1, r2 = ooo.get_active_integrals(ooo.mo_coeff)
    assert abs(c0.item() - float(r0)) < 1e-9
    assert (c1.cpu() - r1).abs().max() < 1e-10 and (c2.cpu() - r2).abs().max() < 1e-10
    # dE/dtheta against central differences of the engine's own energy
    h = 1e-5
    for j in torch.argsort(grad[:pqc.theta_shape].abs(), descending=True)[:2].tolist():
        tp, tm = theta.clone(), theta.clone()
        tp[j] += h
        tm[j] -= h
        fd = (oo.energy_from_parameters(tp).item() - oo.energy_from_parameters(tm).item()) / (2 * h)
        assert abs(fd - grad[j].item()) < 1e-6 * max(1.0, abs(fd))


def test_oo_evaluation_n200_cas66_fused_vs_staged():
    """BASELINE configs[2] as an OO evaluation: N = 200, CAS(6e,6o), n_occ = 20 (M = 26).  The oracle
    is out of reach (3 x 2.56 TFLOP on the CPU); the one-call evaluation (oovqe_cas_eval) is checked
    against the staged kernels (half transform -> finish -> Fock stage, each parity-tested against the
    oracle at smaller N) on identical inputs, plus size-independent properties."""
    import auto_oo_amd as aoo
    from auto_oo_amd import ops
    N, ncas, nelecas, n_occ = 200, 6, 6, 20
    M = n_occ + ncas
    gen = torch.Generator(device=DEV).manual_seed(5)
    B = torch.randn((N, N, N), generator=gen, dtype=torch.float64, device=DEV)
    B = 0.5 * (B + B.transpose(1, 2))
    g = torch.einsum("Lpq,Lrs->pqrs", B[:24], B[:24]) / 24.0          # 8-fold symmetric, 12.8 GB
    del B
    h = torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV)
    h = 0.5 * (h + h.T) / N ** 0.5
    Q, _ = torch.linalg.qr(torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV))
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    theta = torch.tensor(np.random.default_rng(8).uniform(0, 2 * np.pi, pqc.theta_shape))
    g1, g2 = pqc.get_rdms(theta)
    rows, cols = aoo.excitations.tril_tables(N, aoo.non_redundant_indices(
        np.arange(n_occ), n_occ + np.arange(ncas), np.arange(M, N), False))
    kr, kc = torch.as_tensor(rows).to(DEV), torch.as_tensor(cols).to(DEV)
    fused = ops.cas_eval(g, h, Q.contiguous(), g1[None].contiguous(), g2[None].contiguous(), 31.0, n_occ, ncas,
                         kr, kc, want_matrices=True, eri_flags=ops.eri_flags(g))
    # ... and with the tile-packed resident copy of the integrals (stage 1 = half_tiles_kernel)
    tiles = ops.eri_pack(g)
    assert tiles.numel() == 20100 * 91 * 256
    packed = ops.cas_eval(g, h, Q.contiguous(), g1[None].contiguous(), g2[None].contiguous(), 31.0, n_occ, ncas,
                          kr, kc, want_matrices=True, eri_flags=3, g_packed=tiles)
    assert "half_tiles_kernel" in aoo._lib.load().oovqe_last_stage1_kernel().decode()
    del tiles
    for key in ("E", "gvec", "c1", "c2", "fock", "gmat"):
        assert (packed[key] - fused[key]).abs().max() <= 1e-11 * max(1.0, float(fused[key].abs().max())), key
    T2 = ops.cas_half_transform(g, Q.contiguous(), M)
    Gm, hmo = ops.cas_finish_transform(T2, h, Q.contiguous(), M)
    staged = ops.cas_energy_gradient(Gm, hmo, g1[None].contiguous(), g2[None].contiguous(), 31.0, n_occ, ncas,
                                     kr, kc, want_matrices=True)
    scale = float(staged["gvec"].abs().max())
    assert abs(fused["E"].item() - staged["E"].item()) < 1e-9 * max(1.0, abs(staged["E"].item()))
    assert (fused["gvec"] - staged["gvec"]).abs().max() < 1e-10 * max(1.0, scale)
    assert (fused["c2"] - staged["c2"]).abs().max() < 1e-11 and (fused["c1"] - staged["c1"]).abs().max() < 1e-11
    assert (fused["fock"] - staged["fock"]).abs().max() < 1e-10 * max(1.0, float(staged["fock"].abs().max()))
    # orbital-gradient matrix is antisymmetric; virtual rows of the generalized Fock matrix vanish
    assert (fused["gmat"] + fused["gmat"].T).abs().max() < 1e-12 * max(1.0, scale)
    assert fused["fock"][M:].abs().max() == 0.0


@pytest.mark.parametrize("N,n_occ,ncas", [(64, 4, 6), (100, 34, 6), (130, 14, 6), (96, 10, 6), (53, 3, 4)])
def test_cas_eval_large_n_rs_symmetry_reads_the_tile_triangle(N, n_occ, ncas):
    """N > 48 with both symmetry flags: stage 1 reads, for every column tile of a slab, only the row
    tiles up to the diagonal one (54-60 % of the bytes) and symmetrises the slab's result in
    registers.  Same outputs as with the p <-> q flag alone and with no flag (one, two and three
    16-wide tiles of occupied + active orbitals), to rounding -- and the same again from the tile-packed
    resident copy of the integrals (oovqe_eri_pack + oovqe_cas_eval_packed: half_tiles_kernel)."""
    import auto_oo_amd as aoo
    from auto_oo_amd import ops
    M = n_occ + ncas
    gen = torch.Generator(device=DEV).manual_seed(100 + N)
    B = torch.randn((16, N, N), generator=gen, dtype=torch.float64, device=DEV)
    B = 0.5 * (B + B.transpose(1, 2))
    g = (torch.einsum("Lpq,Lrs->pqrs", B, B) / 16.0).contiguous()
    h = torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV)
    h = 0.5 * (h + h.T) / N ** 0.5
    Q, _ = torch.linalg.qr(torch.randn((N, N), generator=gen, dtype=torch.float64, device=DEV))
    rng = np.random.default_rng(N)
    g1 = torch.tensor(rng.standard_normal((1, ncas, ncas))).to(DEV)
    g2 = torch.tensor(rng.standard_normal((1, ncas, ncas, ncas, ncas))).to(DEV)
    rows, cols = aoo.excitations.tril_tables(N, aoo.non_redundant_indices(
        np.arange(n_occ), n_occ + np.arange(ncas), np.arange(M, N), False))
    kr, kc = torch.as_tensor(rows).to(DEV), torch.as_tensor(cols).to(DEV)
    assert ops.eri_flags(g) == 3
    outs = [ops.cas_eval(g, h, Q.contiguous(), g1, g2, 3.0, n_occ, ncas, kr, kc, want_matrices=True,
                         want_integrals=True, eri_flags=f) for f in (0, 1, 3)]
    tiles = ops.eri_pack(g)
    outs.append(ops.cas_eval(g, h, Q.contiguous(), g1, g2, 3.0, n_occ, ncas, kr, kc, want_matrices=True,
                             want_integrals=True, eri_flags=3, g_packed=tiles))
    assert "half_tiles_kernel" in aoo._lib.load().oovqe_last_stage1_kernel().decode()
    for key in ("c0", "c1", "c2", "E", "gvec", "fock", "gmat", "Gm", "hmo"):
        scale = max(1.0, float(outs[0][key].abs().max()))
        for o in outs[1:]:
            assert (outs[0][key] - o[key]).abs().max() <= 2e-12 * scale, key
    # a packed copy without both flags is refused
    with pytest.raises(aoo._lib.OovqeError):
        ops.cas_eval(g, h, Q.contiguous(), g1, g2, 3.0, n_occ, ncas, kr, kc, eri_flags=1, g_packed=tiles)


def test_batched_hessian_and_newton_step_beyond_n48():
    """A stack of geometries beyond N = 48 keeps the tile-packed copy of its integrals (OO_pqc_batch: both flags); the
    one-call energy + gradient + Hessian and the lockstep Newton step run on it and agree with the
    single-geometry objects (which keep their own copy)."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    N, G = 52, 2
    pqc = aoo.Parameterized_circuit(3, 4, None, ansatz="ucc")
    mols, coeffs, objs = [], [], []
    for g in range(G):
        P = synthetic_problem(N, 4100 + g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 16))
        coeffs.append(P["oao_mo_coeff"])
        objs.append(aoo.OO_pqc(pqc, mols[-1], 3, 4, oao_mo_coeff=P["oao_mo_coeff"]))
    batch = aoo.OO_pqc_batch(pqc, mols, 3, 4, oao_mo_coeffs=coeffs)
    assert batch.eri_flags == 3 and batch._eri_packed is not None
    assert batch._eri_packed.shape == (G, N * (N + 1) // 2 * 10 * 256)
    th = torch.tensor(np.random.default_rng(2).uniform(0, 2 * np.pi, (G, 4)), device=DEV)
    E, g, H = batch.energy_gradient_hessian(th)
    for i, oo in enumerate(objs):
        assert oo._eri_packed() is not None
        e1, g1 = oo.energy_and_gradient(th[i])
        h1 = oo.full_hessian(th[i])
        assert abs(E[i].item() - e1.item()) < 1e-11 and (g[i] - g1).abs().max() < 1e-11
        assert (H[i] - h1).abs().max() < 1e-10 * max(1.0, h1.abs().max().item())
    e_before = batch.energy(th)
    new_t, e_new, low = batch.damped_newton_step(th)
    assert (e_new < e_before).all()
    assert (batch.energy(new_t) - e_new).abs().max().item() == 0.0


def test_tile_packed_copy_layout():
    """The format include/oovqe.h documents for N > 48: slabs p <= q; per slab the tiles (R, S), R <= S, column
    tile by column tile; per tile [k-step pair][lane][2] with lane = 16 (row mod 4) + column, k-step = row / 4;
    diagonal tiles halved; zero beyond N."""
    from auto_oo_amd import ops
    N = 53
    gen = torch.Generator(device=DEV).manual_seed(9)
    B = torch.randn((6, N, N), generator=gen, dtype=torch.float64, device=DEV)
    B = 0.5 * (B + B.transpose(1, 2))
    g = (torch.einsum("Lpq,Lrs->pqrs", B, B) / 6.0).contiguous()
    nst = (N + 15) // 16
    tiles = ops.eri_pack(g).reshape(N * (N + 1) // 2, nst * (nst + 1) // 2, 2, 64, 2).cpu()
    gc = g.cpu()
    pad = torch.zeros((16 * nst, 16 * nst), dtype=torch.float64)
    slab = 0
    for p in range(N):
        for q in range(p, N):
            if (p, q) in ((0, 0), (0, 7), (3, 3), (17, 40), (52, 52)):
                pad.zero_()
                pad[:N, :N] = gc[p, q]
                t = 0
                for S in range(nst):
                    for R in range(S + 1):
                        blk = pad[16 * R:16 * R + 16, 16 * S:16 * S + 16] * (0.5 if R == S else 1.0)
                        # blk[4 j + lq, lr] sits at [j // 2][16 lq + lr][j % 2]
                        want = blk.reshape(2, 2, 4, 16).permute(0, 2, 3, 1).reshape(2, 64, 2)
                        assert torch.equal(tiles[slab, t], want), (p, q, R, S)
                        t += 1
            slab += 1


def test_batched_evaluation_beyond_n48_matches_single():
    """Three geometries at N = 96, CAS(6e,6o), M = 16 in one call: the streaming stage 1 with a batch
    dimension and the batched U = C^T T2 contraction (large enough, 3 x 768 strips, for the
    two-strips-per-wave kernel) against the three single evaluations (one-strip kernel) -- both
    symmetry flags and none."""
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    N, ncas, nelecas, nelec, G = 96, 6, 6, 26, 3
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="ucc")
    mols, coeffs = [], []
    for g in range(G):
        P = synthetic_problem(N, 3100 + g)
        mols.append(aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"] + g, nelec))
        coeffs.append(P["oao_mo_coeff"])
    thetas = torch.tensor(np.random.default_rng(6).uniform(0, 2 * np.pi, (G, pqc.theta_shape)))
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=coeffs)
    assert batch.eri_flags == 3
    eg = batch.energy_and_gradient(thetas).cpu()
    batch.eri_flags = 0
    eg0 = batch.energy_and_gradient(thetas).cpu()
    assert (eg - eg0).abs().max() < 1e-10 * max(1.0, float(eg0.abs().max()))
    for g in range(G):
        single = aoo.OO_pqc(pqc, mols[g], ncas, nelecas, oao_mo_coeff=coeffs[g])
        E, grad = single.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - E.item()) < 1e-10
        assert (eg[g, 1:] - grad.cpu()).abs().max() < 1e-10


def test_kupccd_cas88_sector_engine_vs_dense_register_and_oracle():
    """configs[4] at full size against two independent realisations: the sector engine's CAS(8e,8o)
    kUpCCD state (4 900 determinants, scattered to the 16-qubit register) against the dense statevector
    kernel (oovqe_circuit_state, 65 536 amplitudes: another kernel, another data layout) and against the
    oracle's gate-level FermionicDoubleExcitation decompositions (ansatze/kUpCCD.py:94-130); the sector
    RDMs against oovqe_rdms on the dense state (pqc.py:192-218); the reverse-mode theta-gradient against
    forward-mode derivative RDMs of the dense tangent states, for all 56 thetas."""
    import auto_oo_amd as aoo
    from auto_oo_amd import ops
    from oracle import cpu_ref as R
    ncas, nelecas = 8, 8
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=1)
    eng = pqc._sector
    n_theta = int(pqc.theta_shape)
    assert n_theta == 56 and eng.Dc == 4900
    rng = np.random.default_rng(88)
    th = torch.tensor(rng.uniform(0, 2 * np.pi, (1, n_theta)), device=DEV)
    psi_c, psi_sector = eng.state(th, dense=True)
    psi, dpsi = ops.circuit_state(th, pqc._gates_dev, pqc._n_gates, 2 * ncas, pqc._init_index, tangents=True)
    assert psi.shape == (1, 1 << 16) and dpsi.shape == (1, n_theta, 1 << 16)
    assert (psi_sector - psi).abs().max().item() < 1e-13
    # the oracle: gate by gate on the 65 536-vector (complex128, imaginary part = rounding dust)
    ref = R.kupccd_state(th[0].cpu(), ncas, nelecas, 1)
    assert ref.imag.abs().max().item() < 1e-12
    assert (psi_sector[0].cpu() - ref.real).abs().max().item() < 1e-12
    # RDMs
    g1s, g2s = eng.rdms(psi_c)
    g1d, g2d = ops.rdms(psi, psi, ncas)
    assert (g1s - g1d).abs().max().item() < 1e-11
    assert (g2s - g2d).abs().max().item() < 1e-11
    # theta-gradient of c1.gamma + c2.Gamma: adjoint sweep in the sector against dense tangents
    c1 = torch.tensor(rng.standard_normal((ncas, ncas)), device=DEV)
    c2 = torch.tensor(rng.standard_normal((ncas,) * 4), device=DEV)
    dth = eng.adjoint(th, psi_c, c1, c2)[0]
    gam, Gam = ops.rdms_tangent(psi, dpsi, ncas)                   # [1, 57, a, a], [1, 57, a, a, a, a]
    dref = (gam[0, 1:] * c1).sum(dim=(1, 2)) + (Gam[0, 1:] * c2).sum(dim=(1, 2, 3, 4))
    assert dref.shape == (n_theta,)
    assert (dth - dref).abs().max().item() < 1e-10 * max(1.0, dref.abs().max().item())


def _kupccd_problem(N, ncas, nelecas, nelec, seed, k=1):
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    P = synthetic_problem(N, seed)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="kupccd", k=k)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    return P, mol, pqc, oo


def test_sector_second_derivatives_vs_dense_register_cas66():
    """Second derivatives inside the (N_alpha, N_beta) sector (round 4: tangent and second-tangent states from
    oovqe_sector_state_deriv, derivative RDMs by polarisation of the plain RDM kernel, the theta-theta block from
    the operator applied to psi and its first tangents) against the dense-register kernels (second tangents on the 2^12 register, transition RDMs) at
    kUpCCD CAS(6e,6o): derivative RDMs and every block of OO_pqc.full_hessian (oo_pqc.py:103-148)."""
    P, mol, pqc, oo = _kupccd_problem(18, 6, 6, 10, 660)
    assert pqc._use_sector and pqc.n_qubits == 12
    theta = torch.tensor(np.random.default_rng(66).uniform(0, 2 * np.pi, pqc.theta_shape), device=DEV)
    g1s, g2s = pqc.rdms_with_derivatives(theta)
    Hs = oo.full_hessian(theta)
    gs = oo.full_gradient(theta)
    pqc._use_sector = False
    try:
        g1d, g2d = pqc.rdms_with_derivatives(theta)
        Hd = oo.full_hessian(theta)
        gd = oo.full_gradient(theta)
    finally:
        pqc._use_sector = True
    assert g1s.shape == g1d.shape and g2s.shape == g2d.shape
    assert (g1s - g1d).abs().max().item() < 1e-12 and (g2s - g2d).abs().max().item() < 1e-12
    scale = max(1.0, Hd.abs().max().item())
    assert (Hs - Hd).abs().max().item() < 1e-10 * scale
    assert (Hs - Hs.T).abs().max().item() < 1e-10 * scale
    assert (gs - gd).abs().max().item() < 1e-10 * max(1.0, gd.abs().max().item())
    # the theta-theta block two ways inside the sector: 1 + n_theta applications of the operator
    # (oovqe_sector_lambda: H_jk = tau_jk . lam(psi) + tau_j . lam(tau_k), the default) against the polarisation of
    # the quadratic form through 4 n_pairs RDM evaluations; random coefficients without any symmetry
    rng = np.random.default_rng(3)
    c1 = torch.tensor(rng.standard_normal((6, 6)), device=DEV)
    c2 = torch.tensor(rng.standard_normal((6,) * 4), device=DEV)
    th2 = pqc._theta2d(theta)
    Ha = pqc._sector.circuit_hessian(th2, pqc._gates, c1, c2)
    Hb = pqc._sector.circuit_hessian(th2, pqc._gates, c1, c2, by_rdms=True)
    assert (Ha - Hb).abs().max().item() < 1e-10 * max(1.0, Hb.abs().max().item())


def test_sector_second_derivatives_vs_oracle_autograd_cas44():
    """The same against autograd through the oracle's gate-level kUpCCD circuit at CAS(4e,4o) (the sector
    engine forced on an 8-qubit register): OO_pqc.full_hessian == hessian(energy_from_parameters)
    (test/test_oo_pqc.py:101-125 for this ansatz), 1e-8."""
    from oracle import cpu_ref as R
    from torch.autograd.functional import hessian as thessian
    P, mol, pqc, oo = _kupccd_problem(8, 4, 4, 8, 440)
    assert pqc._sector.fits()
    pqc._use_sector = True
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 8)
    ooo = R.OracleOOPQC(R.OraclePQC(4, 4, "kupccd", k=1), omol, 4, 4, P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(44).uniform(0, 2 * np.pi, pqc.theta_shape))
    H = oo.full_hessian(theta.to(DEV)).cpu()
    nt, nk = int(pqc.theta_shape), oo.n_kappa
    # (the blocks the sector engine is behind: theta-theta and kappa-theta; the kappa-kappa block never sees
    # the circuit kernels and is pinned by the dense-register tests)
    htt = thessian(ooo.energy_from_parameters, theta)
    hkt = torch.autograd.functional.jacobian(ooo.orbital_gradient, theta)
    assert (H[:nt, :nt] - htt).abs().max().item() < 1e-8
    assert (H[nt:, :nt] - hkt).abs().max().item() < 1e-8
    assert (oo.full_gradient(theta.to(DEV)).cpu()[:nt]
            - torch.autograd.functional.jacobian(ooo.energy_from_parameters, theta)).abs().max().item() < 1e-8


def _fabric_problem(N, ncas, nelecas, nelec, seed, n_layers=2):
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    P = synthetic_problem(N, seed)
    mol = aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz="np_fabric", n_layers=n_layers)
    oo = aoo.OO_pqc(pqc, mol, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"])
    return P, mol, pqc, oo


def test_sector_second_derivatives_with_shared_parameters_np_fabric_cas66():
    """GateFabric (np_fabric, the ansatz of every reference integration test: test/test_oo_pqc.py:92,167): its
    OrbitalRotation is two Givens rotations on ONE angle (pqc.py:79-83, 136-160), so the tangent states are sums over
    the gates of a parameter and the second tangents sums over gate PAIRS (round 5: SectorEngine.tangent_states).
    CAS(6e,6o), 12 qubits, inside the (N_alpha, N_beta) sector against the dense 2^12 register: derivative RDMs,
    full gradient, every block of OO_pqc.full_hessian (oo_pqc.py:103-148); and a damped Newton step lowers the
    energy (full_optimization's body was out of reach for this ansatz beyond 10 qubits)."""
    import auto_oo_amd as aoo
    P, mol, pqc, oo = _fabric_problem(18, 6, 6, 10, 661)
    assert pqc._use_sector and pqc.n_qubits == 12
    assert max(len(g_) for g_ in pqc._sector.param_gate_lists(pqc._gates)) == 2
    theta = torch.tensor(np.random.default_rng(67).uniform(0, 2 * np.pi, pqc.theta_shape), device=DEV)
    g1s, g2s = pqc.rdms_with_derivatives(theta)
    Hs = oo.full_hessian(theta)
    gs = oo.full_gradient(theta)
    pqc._use_sector = False
    try:
        g1d, g2d = pqc.rdms_with_derivatives(theta)
        Hd = oo.full_hessian(theta)
        gd = oo.full_gradient(theta)
    finally:
        pqc._use_sector = True
    assert (g1s - g1d).abs().max().item() < 1e-12 and (g2s - g2d).abs().max().item() < 1e-12
    scale = max(1.0, Hd.abs().max().item())
    assert (Hs - Hd).abs().max().item() < 1e-10 * scale
    assert (Hs - Hs.T).abs().max().item() < 1e-10 * scale
    assert (gs - gd).abs().max().item() < 1e-10 * max(1.0, gd.abs().max().item())
    th_s = torch.tensor(np.random.default_rng(5).normal(0, 0.3, pqc.theta_shape), device=DEV)
    e0 = oo.energy_from_parameters(th_s).item()
    kap0 = torch.zeros(oo.n_kappa, dtype=torch.float64, device=DEV)
    new, low = aoo.NewtonStep(verbose=0).damped_newton_step(oo.energy_from_parameters, (th_s, kap0),
                                                            oo.full_gradient(th_s), oo.full_hessian(th_s))
    assert oo.energy_from_parameters(new[0], new[1]).item() < e0


def test_sector_second_derivatives_with_shared_parameters_vs_oracle_autograd_cas44():
    """The same against autograd through the oracle's gate-level GateFabric circuit at CAS(4e,4o) (the sector engine
    forced on the 8-qubit register): theta-theta and kappa-theta blocks of OO_pqc.full_hessian and the circuit
    gradient (test/test_oo_pqc.py:101-125), 1e-8."""
    from oracle import cpu_ref as R
    from torch.autograd.functional import hessian as thessian
    P, mol, pqc, oo = _fabric_problem(8, 4, 4, 8, 441)
    assert pqc._sector.fits()
    pqc._use_sector = True
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], 8)
    ooo = R.OracleOOPQC(R.OraclePQC(4, 4, "np_fabric", n_layers=2), omol, 4, 4, P["oao_mo_coeff"])
    theta = torch.tensor(np.random.default_rng(45).uniform(0, 2 * np.pi, pqc.theta_shape))
    H = oo.full_hessian(theta.to(DEV)).cpu()
    nt = int(pqc.theta_shape)
    htt = thessian(ooo.energy_from_parameters, theta)
    hkt = torch.autograd.functional.jacobian(ooo.orbital_gradient, theta)
    assert (H[:nt, :nt] - htt).abs().max().item() < 1e-8
    assert (H[nt:, :nt] - hkt).abs().max().item() < 1e-8
    assert (oo.full_gradient(theta.to(DEV)).cpu()[:nt]
            - torch.autograd.functional.jacobian(ooo.energy_from_parameters, theta)).abs().max().item() < 1e-8


def _sector_batch(N, ncas, nelecas, nelec, G, seed, ansatz="kupccd", **kw):
    import auto_oo_amd as aoo
    from auto_oo_amd.synthetic import synthetic_problem
    pqc = aoo.Parameterized_circuit(ncas, nelecas, None, ansatz=ansatz, **kw)
    probs = [synthetic_problem(N, seed + 1000 * g) for g in range(G)]
    mols = [aoo.Moldata(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec) for P in probs]
    objs = [aoo.OO_pqc(pqc, m, ncas, nelecas, oao_mo_coeff=P["oao_mo_coeff"]) for m, P in zip(mols, probs)]
    batch = aoo.OO_pqc_batch(pqc, mols, ncas, nelecas, oao_mo_coeffs=[P["oao_mo_coeff"] for P in probs])
    return pqc, probs, objs, batch


@pytest.mark.parametrize("ansatz,kw", [("kupccd", {"k": 1}), ("np_fabric", {"n_layers": 2})])
def test_sector_circuits_in_the_geometry_batch_cas44(ansatz, kw):
    """The sector engine composed with the geometry batch (round 5; ansatze/kUpCCD.py:36-154 under oo_pqc.py:64-148):
    OO_pqc_batch on a circuit whose state lives in the (N_alpha, N_beta) sector (forced here on the 8-qubit register
    of CAS(4e,4o)) -- energy + full gradient, full Hessian and a lockstep damped Newton step of 3 geometries against
    the per-geometry OO_pqc objects (1e-11 / 1e-10), against the dense-register batched kernels (the same stack
    with the sector engine switched off) and against the oracle (energy, gradient, theta-theta block: 1e-8)."""
    import auto_oo_amd as aoo
    from oracle import cpu_ref as R
    from torch.autograd.functional import hessian as thessian
    N, ncas, nelecas, nelec, G = 10, 4, 4, 8, 3
    pqc, probs, objs, batch = _sector_batch(N, ncas, nelecas, nelec, G, 7100, ansatz, **kw)
    assert pqc._sector.fits()
    pqc._use_sector = True
    nt = int(pqc.theta_shape)
    rng = np.random.default_rng(71)
    thetas = torch.tensor(rng.uniform(0, 2 * np.pi, (G, nt)), device=DEV)
    eg = batch.energy_and_gradient(thetas)
    E, grad, H = batch.energy_gradient_hessian(thetas)
    for g, oo in enumerate(objs):
        e1, g1 = oo.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - e1.item()) < 1e-11 and (eg[g, 1:] - g1).abs().max().item() < 1e-11
        assert abs(E[g].item() - e1.item()) < 1e-11 and (grad[g] - g1).abs().max().item() < 1e-10
        H1 = oo.full_hessian(thetas[g])
        assert (H[g] - H1).abs().max().item() < 1e-10 * max(1.0, H1.abs().max().item())
    # per-geometry CAS coefficients inside the sector kernels (one launch sequence for all geometries: the default at
    # ncas = 4, 8) against the loop over geometries that other active spaces take
    assert pqc._sector.geometry_coefficients_ok()
    pqc._sector.geometry_coefficients_ok = lambda: False
    try:
        eg_l = batch.energy_and_gradient(thetas)
        E_l, grad_l, H_l = batch.energy_gradient_hessian(thetas)
    finally:
        del pqc._sector.geometry_coefficients_ok
    assert (eg - eg_l).abs().max().item() < 1e-11 and (H - H_l).abs().max().item() < 1e-10
    # the oracle on geometry 0
    P = probs[0]
    omol = R.OracleMol(P["int1e_ao"], P["int2e_ao"], P["overlap"], P["nuc"], nelec)
    ooo = R.OracleOOPQC(R.OraclePQC(ncas, nelecas, ansatz, **kw), omol, ncas, nelecas, P["oao_mo_coeff"])
    th0 = thetas[0].cpu()
    assert abs(ooo.energy_from_parameters(th0).item() - E[0].item()) < 1e-9
    assert (ooo.full_gradient(th0) - grad[0].cpu()).abs().max().item() < 1e-8
    assert (thessian(ooo.energy_from_parameters, th0) - H[0, :nt, :nt].cpu()).abs().max().item() < 1e-8
    # the dense-register batched kernels on the same stack
    pqc._use_sector = False
    try:
        Ed, gd, Hd = batch.energy_gradient_hessian(thetas)
    finally:
        pqc._use_sector = True
    assert (E - Ed).abs().max().item() < 1e-11 and (grad - gd).abs().max().item() < 1e-10
    assert (H - Hd).abs().max().item() < 1e-10 * max(1.0, Hd.abs().max().item())
    # one lockstep damped Newton step == the per-geometry steps (oo_pqc.py:172-196)
    th_s = torch.tensor(rng.normal(0, 0.3, (G, nt)), device=DEV)
    e_before = batch.energy(th_s)
    new_t, e_new, low = batch.damped_newton_step(th_s)
    assert (e_new < e_before).all()
    opt = aoo.NewtonStep(verbose=0)
    for g, oo in enumerate(objs):
        kap0 = torch.zeros(oo.n_kappa, dtype=torch.float64, device=DEV)
        new, lo = opt.damped_newton_step(oo.energy_from_parameters, (th_s[g], kap0), oo.full_gradient(th_s[g]),
                                         oo.full_hessian(th_s[g]))
        assert (new[0] - new_t[g]).abs().max().item() < 1e-8
        assert abs(oo.energy_from_parameters(new[0], new[1]).item() - e_new[g].item()) < 1e-9
        assert abs(float(lo) - low[g].item()) < 1e-8


def test_kupccd_cas88_geometry_batch_equals_single_geometries():
    """configs[4]'s circuit on configs[3]'s loop: kUpCCD CAS(8e,8o) (4 900-determinant sector, 56 thetas) over a stack
    of geometries -- batched energy + gradient and full Hessian equal the single-geometry OO_pqc values."""
    pqc, probs, objs, batch = _sector_batch(20, 8, 8, 16, 2, 8800, "kupccd", k=1)
    assert pqc._use_sector and pqc._sector.Dc == 4900
    nt = int(pqc.theta_shape)
    thetas = torch.tensor(np.random.default_rng(88).uniform(0, 2 * np.pi, (2, nt)), device=DEV)
    eg = batch.energy_and_gradient(thetas)
    E, grad, H = batch.energy_gradient_hessian(thetas)
    for g, oo in enumerate(objs):
        e1, g1 = oo.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - e1.item()) < 1e-10 and (eg[g, 1:] - g1).abs().max().item() < 1e-10
        H1 = oo.full_hessian(thetas[g])
        assert (H[g] - H1).abs().max().item() < 1e-9 * max(1.0, H1.abs().max().item())
        assert (grad[g] - g1).abs().max().item() < 1e-9


def test_np_fabric_cas66_geometry_batch_takes_the_loop_over_geometries():
    """GateFabric CAS(6e,6o) (12 qubits: sector engine; a^2 = 36 is outside the per-geometry-coefficient kernels, so the
    reverse sweep and the theta-theta block loop over the geometries; parameters shared between gates): batched
    energy + gradient and full Hessian equal the single-geometry OO_pqc values, and a lockstep step lowers every
    energy."""
    pqc, probs, objs, batch = _sector_batch(14, 6, 6, 12, 2, 6600, "np_fabric", n_layers=2)
    assert pqc._use_sector and not pqc._sector.geometry_coefficients_ok()
    nt = int(pqc.theta_shape)
    rng = np.random.default_rng(66)
    thetas = torch.tensor(rng.normal(0, 0.4, (2, nt)), device=DEV)
    eg = batch.energy_and_gradient(thetas)
    E, grad, H = batch.energy_gradient_hessian(thetas)
    for g, oo in enumerate(objs):
        e1, g1 = oo.energy_and_gradient(thetas[g])
        assert abs(eg[g, 0].item() - e1.item()) < 1e-11 and (eg[g, 1:] - g1).abs().max().item() < 1e-11
        H1 = oo.full_hessian(thetas[g])
        assert (H[g] - H1).abs().max().item() < 1e-10 * max(1.0, H1.abs().max().item())
        assert (grad[g] - g1).abs().max().item() < 1e-10
    e0 = batch.energy(thetas)
    _, e_new, _ = batch.damped_newton_step(thetas)
    assert (e_new < e0).all()


def test_kupccd_cas88_damped_newton_step_lowers_the_energy():
    """configs[4] has a Newton step: full gradient + full Hessian of kUpCCD CAS(8e,8o), k = 1 (56 thetas, sector
    engine: 4 900 determinants, 1 596 second tangents) and one damped Newton step of OO_pqc.full_optimization's
    body (oo_pqc.py:172-196).  The Hessian is symmetric, its theta-theta block matches central differences of
    the reverse-mode gradient, and the step lowers the energy."""
    import auto_oo_amd as aoo
    P, mol, pqc, oo = _kupccd_problem(20, 8, 8, 12, 880)
    assert pqc._use_sector and pqc._sector.Dc == 4900
    rng = np.random.default_rng(8)
    theta = torch.tensor(rng.normal(0, 0.3, pqc.theta_shape), device=DEV)
    E0 = oo.energy_from_parameters(theta).item()
    g = oo.full_gradient(theta)
    H = oo.full_hessian(theta)
    nt, n = int(pqc.theta_shape), g.numel()
    assert H.shape == (n, n) and nt == 56
    assert (H - H.T).abs().max().item() < 1e-9 * max(1.0, H.abs().max().item())
    for j in (0, 17, 55):
        e = torch.zeros(nt, dtype=torch.float64, device=DEV)
        e[j] = 1e-5
        fd = (oo.full_gradient(theta + e) - oo.full_gradient(theta - e)) / 2e-5
        assert (fd - H[:, j]).abs().max().item() < 1e-5 * max(1.0, H[:, j].abs().max().item())
    kappa = torch.zeros(oo.n_kappa, dtype=torch.float64, device=DEV)
    new, low = aoo.NewtonStep(verbose=0).damped_newton_step(oo.energy_from_parameters, (theta, kappa), g, H)
    E1 = oo.energy_from_parameters(new[0], new[1]).item()
    assert E1 < E0 - 1e-6, (E0, E1)


def test_sector_second_order_autodiff_at_nonzero_kappa_equals_dense_register():
    """hessian(energy_from_parameters)(theta, kappa) at kappa != 0 (oo_pqc.py:103-125 differentiates at any
    point) on a sector circuit (kUpCCD CAS(6e,6o), 12 qubits): the sector engine's second derivatives behind
    torch autodiff against the dense-register kernels behind the same rules."""
    from torch.autograd.functional import hessian as thessian
    P, mol, pqc, oo = _kupccd_problem(16, 6, 6, 10, 661)
    assert pqc._use_sector
    rng = np.random.default_rng(67)
    theta = torch.tensor(rng.uniform(0, 2 * np.pi, pqc.theta_shape))
    kappa = torch.tensor(rng.normal(0, 0.05, oo.n_kappa))
    hs = thessian(oo.energy_from_parameters, (theta, kappa))
    pqc._use_sector = False
    try:
        hd = thessian(oo.energy_from_parameters, (theta, kappa))
    finally:
        pqc._use_sector = True
    for i in range(2):
        for j in range(2):
            scale = max(1.0, hd[i][j].abs().max().item())
            assert (hs[i][j] - hd[i][j]).abs().max().item() < 1e-9 * scale, (i, j)
