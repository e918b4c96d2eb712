/* oovqe.h -- C ABI of liboovqe_hip.so, the MI355X (gfx950) engine behind auto_oo's
 * OO_energy / OO_pqc / Parameterized_circuit cost-function API.
 *
 * The reference (Emieeel/auto_oo) has no FFI layer: its hot path is a sequence of
 * pennylane.math / torch / PennyLane-simulator calls.  Each entry point below replaces one such
 * call sequence; the reference file:line it stands in for is cited per function.  Paths are
 * relative to the reference repository root.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch.Tensor.data_ptr()), caller-owned;
 *   - all floating point data is fp64, C-contiguous (row-major), exactly the reference's layout;
 *   - index tables are int32;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *   - return value: 0 on success, negative on error (oovqe_last_error() gives the text);
 *     no exceptions cross the ABI, no global state besides the last-error string.
 */
#ifndef OOVQE_H
#define OOVQE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* oovqe_stream_t;

/* ---- library ------------------------------------------------------------------------------ */
int         oovqe_version(void);
const char* oovqe_last_error(void);
/* number of visible HIP devices (<=0: none / runtime error) */
int         oovqe_device_count(void);

/* HIP-event timing of the dominant kernel (the N^4 half-transform sweep) on its launch stream:
 * between begin and end every oovqe_cas_half_transform launch is bracketed by two events;
 * end() returns the summed kernel time and the number of launches.  Used by bench.py only. */
int         oovqe_profile_begin(void);
int         oovqe_profile_end(double* total_ms, int* count);
/* bracket EVERY launch of an evaluation (costs ~8 us of dispatch gap per bracketed launch) */
int         oovqe_profile_begin_detail(void);
/* the same, broken down by launch of oovqe_oo_eval[_batch]: label 0 = half-transform,
 * 1 = circuit + RDMs (absent when they ride along launch 2), 2 = p->n contraction,
 * 3 = Fock-column (panel) kernel, 4 = final assembly */
int         oovqe_profile_end_labels(double* ms_by_label, int* count_by_label, int n_labels);
/* name and template arguments of the stage-1 (N^4 pass) kernel the last evaluation dispatched, e.g.
 * "half_tri_reg_kernel<11,3,8,3>" ("" before the first evaluation); bench.py's roofline quotes it */
const char* oovqe_last_stage1_kernel(void);

/* ---- gate table for the statevector kernels ----------------------------------------------- *
 * One entry per excitation gate (qml.FermionicDoubleExcitation / FermionicSingleExcitation /
 * DoubleExcitation closed forms; src/auto_oo/ansatze/uccd.py:105-114, kUpCCD.py:118-130).
 * Qubit (wire) w of an n-qubit register is bit (n-1-w) of the basis index (wire 0 = MSB).
 * For every basis index x with (x & (mask_hi|mask_lo)) == mask_hi and partner y = x ^ (mask_hi|mask_lo):
 *     pi = (-1)^popcount(x & mask_par);  c = cos(sign*theta/2);  s = sin(sign*theta/2)
 *     psi'[x] =  c psi[x] + pi s psi[y]
 *     psi'[y] = -pi s psi[x] + c psi[y]
 */
typedef struct {
    uint32_t mask_hi;    /* bits set in x (the "occupied" pair s,r of an FDE)                 */
    uint32_t mask_lo;    /* bits set in y (q,p)                                                */
    uint32_t mask_par;   /* bits whose parity gives the fermionic sign                         */
    int32_t  theta_idx;  /* which parameter drives this gate (<0: fixed angle 0)               */
    int32_t  sign;       /* +1 / -1 multiplies theta                                           */
    int32_t  nfix;       /* popcount(mask_hi|mask_lo): 4 (double) or 2 (single)                */
    int32_t  pos[4];     /* bit positions of mask_hi|mask_lo, ascending                        */
} oovqe_gate_t;

/* Test / measurement switches between realisations that compute the same numbers (which kernel
 * variant a call takes): "half_stream_old", "gm_two_per_cu", "gm_one_per_cu", "fused_chunks" (int),
 * "tri_plain_w", "cas_unfused", "sym_no_rs", "sym_mirror", "sym_simple", "sym_two_step", "no_ride",
 * "tri_mode" (int), "k1_no_pair", "k1_force_wide", "gm_plain_grid", "newton_one_wg", "sector_unfused",
 * "sector_probe" (int; timing only), "hess_vk_pass", "hess_own_stage1", "panel_rows" (int), "k1_force_nt" (int),
 * "newton_no_chol", "tiles_variant" (int), "sector_lambda_w" (int), "sector_rdm_r3" (int), "gm_three_per_cu";
 * "no_ride": 1 = the circuit + RDM step as a launch of its own, 2 = riding on the launch in front of the Fock stage
 * whatever the batch (0: the library decides by the grid of that launch).
 * All 0 by default; the library never reads environment variables.  tests/ and tools/ only. */
int oovqe_debug_set_option(const char* name, int value);
int oovqe_debug_get_option(const char* name);

/* ---- a1: four-index transform --------------------------------------------------------------
 * replaces general_4index_transform / uniform_4index_transform / int2e_transform
 * (src/auto_oo/oo_energy.py:21-30,33-41,49-51):
 *     out[i,j,k,l] = sum_pqrs C0[p,i] C1[q,j] C2[r,k] C3[s,l] M[p,q,r,s]
 * M, out: [N,N,N,N]; C*: [N,N]; work: [N^4] scratch.  out may not alias M or work. */
int oovqe_general_4index_transform(const double* M, const double* C0, const double* C1,
                                   const double* C2, const double* C3, int N,
                                   double* out, double* work, oovqe_stream_t stream);

/* ---- a2/a6: small dense products -----------------------------------------------------------
 * out[m,n] = sum_k A[m,k] B[k,n]         (A: [M,K], B: [K,N])    `@` in oo_energy.py:46,176,201,235
 * out[m,n] = sum_k A[k,m] B[k,n]         (A: [K,M], B: [K,N])    `.T @` in oo_energy.py:46      */
int oovqe_matmul_nn(const double* A, const double* B, int M, int K, int N, double* out,
                    oovqe_stream_t stream);
int oovqe_matmul_tn(const double* A, const double* B, int M, int K, int N, double* out,
                    oovqe_stream_t stream);
/* out[b] = A[b] B[b], b < batch, in one launch (A: [batch,M,K], B: [batch,K,N]): `mo_coeff = oao_coeff @
 * oao_mo_coeff` (oo_energy.py:173-176) for every geometry of a stack */
int oovqe_matmul_nn_batch(const double* A, const double* B, int M, int K, int N, int batch, double* out,
                          oovqe_stream_t stream);

/* generic mode contraction used by all of the above (and exported for tests):
 *   last == 0: out[a,j,b] = sum_k Cm[k*ldc + j] * T[a,k,b]     T: [A,K,B], out: [A,J,B]
 *   last != 0: out[a,j]   = sum_k T[a,k] * Cm[k*ldc + j]       T: [A,K],   out: [A,J]        */
int oovqe_mode_contract(const double* T, const double* Cm, double* out, int64_t A, int K, int J,
                        int64_t B, int ldc, int last, oovqe_stream_t stream);

/* ---- a3/a5: kappa -> U = expm(-K) ----------------------------------------------------------
 * replaces OO_energy.kappa_vector_to_matrix + vector_to_skew_symmetric + math.expm(-K)
 * (src/auto_oo/oo_energy.py:63-87,213-219,226-230).  kap_row/kap_col [n_kappa]: (row,col) with
 * row>col of each non-redundant parameter (np.tril_indices order filtered by params_idx).
 * K: [N,N] out (skew matrix, optional: may be NULL); U: [N,N] out; work: [7*N*N] scratch (only
 * touched when N > 48; below that the whole computation lives in one workgroup's LDS). */
int oovqe_expm_skew(const double* kappa, const int32_t* kap_row, const int32_t* kap_col,
                    int n_kappa, int N, double* K, double* U, double* work,
                    oovqe_stream_t stream);
/* U = expm(sign * X) for a general [N,N] matrix X (sign = -1 reproduces math.expm(-X));
 * work: [6*N*N] scratch (only touched when N > 48). */
int oovqe_expm(const double* X, double sign, int N, double* U, double* work,
               oovqe_stream_t stream);

/* ---- a9/a10/a11: statevector + RDMs --------------------------------------------------------
 * replaces Parameterized_circuit.qnode / uccd_state (src/auto_oo/pqc.py:69-76,121-134,165-172;
 * ansatze/uccd.py:105-114; ansatze/kUpCCD.py:118-130) and get_rdms_from_state (pqc.py:192-221)
 * with E_pq / e_pqrs of utils/active_space.py:29-83 (spin orbital 2p = alpha, 2p+1 = beta).
 *
 * theta [batch, n_theta]; psi out [batch, D] (real amplitudes; D = 2^n_qubits);
 * dpsi out [batch, n_theta, D] tangents d psi / d theta_k, or NULL.                          */
int oovqe_circuit_state(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                        int n_qubits, uint32_t init_index, int batch, double* psi, double* dpsi,
                        oovqe_stream_t stream);
/* gamma[b,p,q] = bra_b^T E_pq ket_b ; Gamma[b,p,q,r,s] = bra_b^T (E_pq E_rs - d_qr E_ps) ket_b.
 * bra == ket gives the reference's RDMs; bra != ket gives transition RDMs (used for
 * d gamma / d theta = T(dpsi,psi) + T(psi,dpsi)).  work: [batch * 2 * ncas^2 * D] scratch.    */
int oovqe_rdms(const double* bra, const double* ket, int n_qubits, int ncas, int batch,
               double* gamma, double* Gamma, double* work, oovqe_stream_t stream);

/* RDMs of psi and their theta-derivatives in one call (what torch autograd produces for the
 * reference when it differentiates get_rdms through the simulator, oo_pqc.py:86-95,113-119):
 *   set 0: gamma, Gamma of psi;  set k>=1: d gamma / d theta_k, d Gamma / d theta_k from the
 *   tangent dpsi[:,k-1,:].  gamma [batch, 1+n_tan, a, a]; Gamma [batch, 1+n_tan, a,a,a,a];
 *   work: [batch * (1+n_tan) * ncas^2 * D] scratch.                                             */
int oovqe_rdms_tangent(const double* psi, const double* dpsi, int n_qubits, int ncas, int n_tan,
                       int batch, double* gamma, double* Gamma, double* work,
                       oovqe_stream_t stream);

/* State, tangents and all RDM sets in one call (= Parameterized_circuit.get_rdms and its
 * theta-jacobian, pqc.py:220-221 under oo_pqc.py:86-95).  For small active spaces (n_qubits <= 10
 * and everything fits 150 KiB of LDS, e.g. CAS(4e,3o)) this is ONE launch of one workgroup per
 * batch element; otherwise it chains oovqe_circuit_state + oovqe_rdms_tangent.
 * psi [batch,D] / dpsi [batch,n_theta,D] may be NULL on the small path (not written back then);
 * work as for oovqe_rdms_tangent (may be NULL on the small path). */
int oovqe_circuit_rdms(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                       int n_qubits, int ncas, uint32_t init_index, int want_tangents, int batch,
                       double* psi, double* dpsi, double* gamma, double* Gamma, double* work,
                       oovqe_stream_t stream);

/* Spin-orbital (unrestricted) RDMs, the restricted=False branch of
 * Parameterized_circuit.get_rdms_from_state (src/auto_oo/pqc.py:192-218, operators
 * utils/active_space.py:29-83): gamma [batch, n, n] = <a+_p a_q>, Gamma [batch, n, n, n, n] =
 * <a+_p a+_q a_r a_s>, n = n_qubits spin orbitals, bilinear in (bra, ket) [batch, 2^n]. */
int oovqe_spin_rdms(const double* bra, const double* ket, int n_qubits, int batch, double* gamma,
                    double* Gamma, oovqe_stream_t stream);

/* ---- a7/a8/a12/a13/a14: fused CAS energy + gradients ---------------------------------------
 * replaces int1e_transform + int2e_transform + molecular_hamiltonian_coefficients
 * (oo_energy.py:204-211; utils/active_space.py:111-212), the energy contraction
 * (oo_energy.py:195-197), fock_core / fock_active / fock_generalized /
 * analytic_gradient_from_integrals (oo_energy.py:238-309) and kappa_matrix_to_vector
 * (oo_energy.py:221-224) WITHOUT materialising the N^4 MO tensor: only
 * Gm[n,x,y,z] = g_mo[n,x,y,z], x,y,z < n_occ+ncas, is formed.
 *
 * eri_flags (oovqe_cas_eval / oovqe_oo_eval / oovqe_oo_eval_batch): properties of g_ao the caller
 * vouches for.  OOVQE_ERI_PQ_SYMMETRIC: g_ao[p,q,:,:] == g_ao[q,p,:,:] EXACTLY (true for the
 * integrals of real orbitals as PySCF's mol.intor('int2e') returns them, which is what
 * Moldata_pyscf.int2e_ao holds, src/auto_oo/moldata_pyscf.py:31; verify with
 * oovqe_eri_symmetry_flags).  The N^4 pass then reads only the N(N+1)/2 slabs p <= q: half the
 * HBM traffic, identical results.  With eri_flags == 0 nothing is assumed about g_ao.
 *
 * Stage 1 (the N^4 pass):  T2[p,q,y,z] = sum_rs C[r,y] g_ao[p,q,r,s] C[s,z],  y,z < M.       */
#define OOVQE_ERI_PQ_SYMMETRIC 1u
/* OOVQE_ERI_RS_SYMMETRIC: g_ao[p,q,r,s] == g_ao[p,q,s,r] EXACTLY (equally true of PySCF's int2e).
 * The half-transformed slabs are then symmetric in (y,z); together with OOVQE_ERI_PQ_SYMMETRIC the
 * batched path stores and contracts only the columns y <= z. */
#define OOVQE_ERI_RS_SYMMETRIC 2u
/* *eri_flags = the OOVQE_ERI_* bits that hold bit for bit for every geometry of the stack g_ao
 * [batch][N^4] (one pass over the tensor; synchronises `stream`). */
int oovqe_eri_symmetry_flags(const double* g_ao, int N, int batch, unsigned* eri_flags,
                             oovqe_stream_t stream);
/* Packed resident copy of integrals that carry BOTH flags (g_packed of oovqe_oo_eval_batch /
 * oovqe_oo_hessian_batch / oovqe_cas_eval_packed).  N <= 48: per geometry the slabs p <= q, of each slab the
 * upper triangle with the diagonal halved (row r = its columns (r & ~1) .. N-1, 0 left of the diagonal, rows
 * back to back) -- 27 % of the tensor at N = 43; the batched N^4 pass then streams this copy instead of
 * picking cache-line fragments out of g_ao.  N > 48: per geometry the slabs p <= q, of each slab the 16 x 16
 * tiles (R, S), R <= S, column tile by column tile, every tile as the four matrix-core operand fragments it
 * is consumed as, diagonal tiles halved, zero beyond N -- 29 % of the tensor at N = 200
 * (half_tiles_kernel).  oovqe_eri_packed_size: doubles per geometry.  g_ao itself stays an argument of
 * every entry point (paths that cannot use the copy read it). */
int64_t oovqe_eri_packed_size(int N);
int oovqe_eri_pack(const double* g_ao, int N, int batch, double* packed, oovqe_stream_t stream);
/* INGEST of a stack of AO tensors (what `OO_energy.__init__` does once per molecule with `mol.int2e_ao`,
 * oo_energy.py:143-171, and the Berry-phase notebook once per loop point): eri_flags[g] (HOST array [batch]) = the
 * OOVQE_ERI_* bits of geometry g, and -- `packed` non-null -- the packed copy of every geometry (meaningful for those
 * that carry both bits).  N <= 48: ONE pass over g_ao (every slab read once, the copy written: 1.27 x the tensor's
 * bytes; the two calls above move 2.3 x); beyond: the check pass and the tile pack.  Synchronises `stream`. */
int oovqe_eri_ingest(const double* g_ao, int N, int batch, double* packed, unsigned* eri_flags,
                     oovqe_stream_t stream);
/* oovqe_cas_eval (below) with the packed copy of g_ao: the same outputs; stage 1 streams g_packed where a
 * kernel for it exists (N > 48: the T2 paths), g_ao otherwise.  eri_flags must carry both bits. */
int oovqe_cas_eval_packed(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                          const double* Gamma, int nrdm, double nuc, int N, int n_occ, int ncas,
                          const int32_t* kap_row, const int32_t* kap_col, int n_kappa, double* work,
                          double* c0, double* c1, double* c2, double* E, double* gvec, double* dE,
                          double* fock, double* gmat, double* Gm, double* hmo, unsigned eri_flags,
                          const double* g_packed, oovqe_stream_t stream);
int oovqe_cas_half_transform(const double* g_ao, const double* C, int N, int M, double* T2,
                             oovqe_stream_t stream);
/* Stage 2: Gm[n,x,y,z] = sum_pq C[p,n] C[q,x] T2[p,q,y,z]; hmo[n,x] = (C^T h_ao C)[n,x].
 * work: [N*M*M*M + N*M] scratch. */
int oovqe_cas_finish_transform(const double* T2, const double* h_ao, const double* C, int N,
                               int M, double* Gm, double* hmo, double* work,
                               oovqe_stream_t stream);
/* Stage 3: everything that is O(N M^3).  nrdm >= 1 RDM sets: set 0 is (gamma, Gamma) itself and
 * yields c0,c1,c2,E, the generalized Fock matrix and the orbital gradient; sets k>=1 are
 * derivative RDMs (d gamma/d theta_k, d Gamma/d theta_k) and yield dE/dtheta_k and the
 * orbital-circuit Hessian column d G_kappa / d theta_k (oo_pqc.py:86-95,113-119).
 * outputs: c0[1], c1[a,a], c2[a^4], E[1], fock[N,N], gmat[N,N] (=2(F-F^T)),
 *          gvec[nrdm, n_kappa] (row 0 = orbital gradient), dE[nrdm-1] (may be NULL if nrdm==1). */
int oovqe_cas_energy_gradient(const double* Gm, const double* hmo, const double* gamma,
                              const double* Gamma, int nrdm, double nuc, int N, int n_occ,
                              int ncas, const int32_t* kap_row, const int32_t* kap_col,
                              int n_kappa, double* c0, double* c1, double* c2, double* E,
                              double* fock, double* gmat, double* gvec, double* dE,
                              oovqe_stream_t stream);
/* The same stage spread over the CUs (a workgroup per (general index, RDM set), then one assembly
 * launch) instead of one workgroup: same arguments plus work, oovqe_cas_energy_gradient_work_size()
 * doubles of scratch (falls back to the one-workgroup kernel when the slices of g_mo[n] it gathers
 * exceed 150 KB of LDS or n_occ + ncas > 64). */
int oovqe_cas_energy_gradient_ws(const double* Gm, const double* hmo, const double* gamma,
                                 const double* Gamma, int nrdm, double nuc, int N, int n_occ,
                                 int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                 int n_kappa, double* c0, double* c1, double* c2, double* E,
                                 double* fock, double* gmat, double* gvec, double* dE, double* work,
                                 oovqe_stream_t stream);
int64_t oovqe_cas_energy_gradient_work_size(int N, int n_occ, int ncas, int nrdm);

/* The whole CAS path in one call: 4 launches (stage 1 [+ q->x inside the same persistent kernel
 * when the sweep is bandwidth-bound], contraction p->n, Fock-column kernel, final assembly); the
 * N^2 M^2 half-transformed tensor stays on chip on the batched path.
 * Same inputs/outputs as the three stages above; work: oovqe_cas_eval_work_size() doubles;
 * fock, gmat, Gm [N,M,M,M], hmo [N,M], dE may be NULL (dE only when nrdm == 1).
 * This is what OO_pqc.energy_from_parameters / full_gradient / orbital_circuit_hessian run on
 * (src/auto_oo/oo_pqc.py:64-134). */
int oovqe_cas_eval(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                   const double* Gamma, int nrdm, double nuc, int N, int n_occ, int ncas,
                   const int32_t* kap_row, const int32_t* kap_col, int n_kappa, double* work,
                   double* c0, double* c1, double* c2, double* E, double* gvec, double* dE,
                   double* fock, double* gmat, double* Gm, double* hmo, unsigned eri_flags,
                   oovqe_stream_t stream);
int64_t oovqe_cas_eval_work_size(int N, int n_occ, int ncas, int nrdm);

/* Inactive / active Fock matrices from FULL MO integrals h_mo [N,N], g_mo [N,N,N,N]: the public
 * helpers OO_energy.fock_core / fock_active (src/auto_oo/oo_energy.py:272-298).  Either output may
 * be NULL. */
int oovqe_fock_core_active(const double* h_mo, const double* g_mo, const double* gamma, int N, int n_occ,
                           int ncas, double* fock_core, double* fock_active, oovqe_stream_t stream);

/* ---- a15: orbital-orbital Hessian -----------------------------------------------------------------
 * replaces full_rdms / y_matrix / analytic_hessian_from_integrals / full_hessian_to_matrix
 * (src/auto_oo/oo_energy.py:311-402) and OO_pqc.orbital_orbital_hessian (oo_pqc.py:127-130).
 * fock [N,N] = generalized Fock matrix (output of oovqe_cas_eval for the same RDMs).
 * H_matrix [n_kappa, n_kappa] (non-redundant pairs) and/or H_full [N,N,N,N]; either may be NULL.
 * work: oovqe_orbital_hessian_work_size() doubles. */
int oovqe_orbital_hessian(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                          const double* Gamma, const double* fock, int N, int n_occ, int ncas,
                          const int32_t* kap_row, const int32_t* kap_col, int n_kappa, double* work,
                          double* H_matrix, double* H_full, oovqe_stream_t stream);
int64_t oovqe_orbital_hessian_work_size(int N, int n_occ, int ncas);

/* The same for a stack of `batch` geometries of identical shape in one call (the geometry index is a
 * grid dimension of every launch): g_ao [batch][N^4], h_ao / C / fock [batch][N^2], gamma [batch][a^2],
 * Gamma [batch][a^4], H_matrix [batch][n_kappa][n_kappa]; work: batch * oovqe_orbital_hessian_work_size()
 * doubles; eri_flags as for oovqe_cas_eval (OOVQE_ERI_PQ_SYMMETRIC: the J-type half transform reads the
 * slabs p <= q only). */
int oovqe_orbital_hessian_batch(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                                const double* Gamma, const double* fock, int N, int n_occ, int ncas,
                                const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int batch,
                                double* work, double* H_matrix, unsigned eri_flags, oovqe_stream_t stream);

/* ---- a16: circuit-circuit Hessian pieces (oo_pqc.py:103-111) ------------------------------------
 * second tangents d^2 psi/d theta_j d theta_k for the listed (j,k) pairs: out [n_pairs, D],
 * scratch [n_pairs, D]; pairs int32 [n_pairs, 2]. */
int oovqe_circuit_second_tangents(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                  int n_gates, int n_qubits, uint32_t init_index,
                                  const int32_t* pairs, int n_pairs, double* out, double* scratch,
                                  oovqe_stream_t stream);
/* H[j,k] = H[k,j] = sum_{set<4} c1 . gamma[pair,set] + c2 . Gamma[pair,set], the four sets being the
 * transition RDMs T(psi_jk,psi), T(psi_j,psi_k), T(psi_k,psi_j), T(psi,psi_jk). */
int oovqe_circuit_hessian_assemble(const double* gamma, const double* Gamma, const double* c1,
                                   const double* c2, int ncas, const int32_t* pairs, int n_pairs,
                                   int n_theta, double* H, oovqe_stream_t stream);

/* ---- f1: damped-Newton direction (the caller of the path) --------------------------------------
 * replaces NewtonStep.newton_step (src/auto_oo/utils/newton_raphson.py:78-129): lowest eigenvalue
 * of the (n_theta + n_kappa)^2 Hessian, level shift nu = mu + rho |lambda_low| when
 * lambda_low < lambda_min and aug != 0 (the reference's augmented Hessian, :107-120), and
 * dp = -(H + nu I)^-1 g (:121-128), for `batch` independent problems.
 * hessian [batch,n,n] (symmetric, lower triangle read, not modified), gradient [batch,n], dp [batch,n],
 * lowest_eigenvalue [batch], shift [batch] (nu; may be NULL); n <= oovqe_newton_direction_max_n();
 * work: oovqe_newton_direction_work_size(n, batch) doubles.
 * One call = the positive-definite fast path below (aug != 0, n <= oovqe_newton_direction_pd_max_n()) followed
 * by the band-reduction route, which then only adds the lowest eigenvalue of the problems the fast path
 * served and does everything for the others.  Failures are loud: dp = NaN (see `info` below). */
int oovqe_newton_direction(const double* hessian, const double* gradient, int n, int batch,
                           double lambda_min, double mu, double rho, int aug, double* work,
                           double* dp, double* lowest_eigenvalue, double* shift,
                           oovqe_stream_t stream);
int64_t oovqe_newton_direction_work_size(int n, int batch);
int oovqe_newton_direction_max_n(void);
/* The same in two calls, so that the caller can keep the eigenvalue of a positive definite Hessian -- a
 * reported number the line search never reads (newton_raphson.py:105-128: dp depends on lambda_low only when
 * lambda_low < lambda_min) -- off the critical path, e.g. on a second stream:
 *
 * oovqe_newton_direction_pd: blocked Cholesky of H (and of H - lambda_min I, the test "the reference would not
 * shift"), one workgroup per factorisation, no workgroup ever waits for another.  info[b] = 1: dp[b] =
 * -H^-1 g and shift[b] = 0 are final; info[b] = 0: not positive definite above lambda_min, dp[b] untouched.
 * work: oovqe_newton_direction_pd_work_size(n, batch) doubles.  n <= oovqe_newton_direction_pd_max_n().
 *
 * oovqe_newton_direction_rest: the band-reduction route (lowest eigenvalue by multisection, level shift, band
 * solve) under the fast path's verdict.  info may be NULL (every problem in full).  which = 0: every problem
 * (lowest eigenvalue only where info[b] == 1); 1: only the problems with info[b] != 1 (the others are not
 * touched: `lowest_eigenvalue` keeps its content); 2: only the lowest eigenvalue of the problems with
 * info[b] == 1.  max_wg: 0, or an upper bound on the workgroups per problem (1: no workgroup waits for
 * another -- the retry after a timed-out hand-off).  On return info[b] (when given) is 1 / 0 as before,
 * 2 = direction from the fast path but its eigenvalue could not be computed (NaN), -1 = a hand-off between
 * workgroups timed out (their co-residency was not granted: dp, lowest, shift = NaN; repeat with max_wg = 1),
 * -2 = aug == 0 (or lambda_min <= 0) and an indefinite Hessian beyond the pivoted one-workgroup kernel
 * (n > 480): dp = NaN, lowest_eigenvalue valid; -3 = the Hessian holds a NaN or an Inf (all outputs NaN).
 * work: oovqe_newton_direction_rest_work_size(n, batch) doubles (its own: not shared with a concurrent call). */
int oovqe_newton_direction_pd(const double* hessian, const double* gradient, int n, int batch,
                              double lambda_min, double* work, double* dp, double* shift, double* info,
                              oovqe_stream_t stream);
int64_t oovqe_newton_direction_pd_work_size(int n, int batch);
int oovqe_newton_direction_pd_max_n(void);
/* 1 when oovqe_newton_direction(n, aug) itself runs the fast path in front of the band route */
int oovqe_newton_direction_has_pd(int n, int aug);
int oovqe_newton_direction_rest(const double* hessian, const double* gradient, int n, int batch,
                                double lambda_min, double mu, double rho, int aug, double* info, int which,
                                int max_wg, double* work, double* dp, double* lowest_eigenvalue,
                                double* shift, oovqe_stream_t stream);
int64_t oovqe_newton_direction_rest_work_size(int n, int batch);

/* Book-keeping of the backtracking line search for `batch` problems in lockstep (NewtonStep.backtracking,
 * src/auto_oo/utils/newton_raphson.py:131-192, per problem).
 * oovqe_linesearch_points: points = flat + t[b] dp, the first n_a entries of a problem to points_a [batch,n_a],
 * the rest to points_b [batch,n-n_a] (n_a = 0 or n: everything to points_a); slope [batch] (may be NULL) =
 * alpha <grad, dp> (newton_raphson.py:12-13).
 * oovqe_linesearch_update: trial [batch] (stride `trial_stride` doubles) = the energies at the points; a problem
 * still searching passes when trial <= energy + t slope (:146-147; a NaN never passes): then best[b] = trial;
 * otherwise t[b] *= beta (:162) and active[b] = 1.  first != 0: every problem is searching.  give_up != 0
 * (after lmax + 1 reductions, :177-183): the problems still searching keep their old parameters (t = 0,
 * best = energy); trial is not read.  flags [4] = {any problem still searching, min of info (0 without),
 * any NaN slope, any searching problem whose slope is not negative (:158: not a descent direction)}. */
int oovqe_linesearch_points(const double* flat, const double* dp, const double* t, const double* grad,
                            double alpha, int n, int n_a, int batch, double* points_a, double* points_b,
                            double* slope, oovqe_stream_t stream);
int oovqe_linesearch_update(const double* trial, int64_t trial_stride, const double* energy, const double* slope,
                            const double* info, double beta, int first, int give_up, int batch, double* t,
                            double* active, double* best, double* flags, oovqe_stream_t stream);

/* The theta-theta block in one call: the five launches above chained (state + tangents, second
 * tangents, operand lists, transition RDMs, contraction).  pairs [n_pairs][2] with j <= k; H
 * [n_theta, n_theta]; work: oovqe_circuit_hessian_work_size() doubles. */
int oovqe_circuit_hessian(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                          int n_qubits, int ncas, uint32_t init_index, const double* c1,
                          const double* c2, const int32_t* pairs, int n_pairs, double* work,
                          double* H, oovqe_stream_t stream);
int64_t oovqe_circuit_hessian_work_size(int n_theta, int n_qubits, int ncas, int n_pairs);
/* The same for `batch` parameter sets in one call: theta [batch][n_theta], c1 [batch][a^2],
 * c2 [batch][a^4], H [batch][n_theta][n_theta]; work: batch * oovqe_circuit_hessian_work_size();
 * batch * 4 * n_pairs <= 65535. */
int oovqe_circuit_hessian_batch(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                                int n_qubits, int ncas, uint32_t init_index, const double* c1,
                                const double* c2, const int32_t* pairs, int n_pairs, int batch,
                                double* work, double* H, oovqe_stream_t stream);

/* ---- a10/a11/a13 at scale: particle-number-sector engine with reverse-mode gradients -------------
 * For circuits that conserve (N_alpha, N_beta) -- UCCD, UCCSD, kUpCCD -- the state lives in a
 * sector of C(a,N_alpha)*C(a,N_beta) determinants (4 900 of 65 536 for CAS(8e,8o)); the sector
 * vector fits one workgroup's LDS, so the whole circuit is ONE launch per batch.
 * Sector tables (device): unrank_a [na] / unrank_b [nb] = occupation strings (orbital p at bit
 * ncas-1-p), rank_a / rank_b [2^ncas] = string -> index or -1.  Compressed index c = ia*nb + ib.
 *   oovqe_sector_state   : theta [batch,n_theta] -> psi_c [batch, na*nb] (and, if not NULL, the
 *                          dense psi [batch, 2^n]: Parameterized_circuit.qnode, pqc.py:165-172)
 *   oovqe_sector_rdms    : gamma [batch,a,a], Gamma [batch,a,a,a,a] (pqc.py:192-221); MFMA Gram
 *   oovqe_sector_adjoint : dtheta [batch,n_theta] = d/dtheta (c1.gamma + c2.Gamma): the reverse
 *                          sweep torch autograd performs for the reference (oo_pqc.py:86-95);
 *                          (round 3: independent of oovqe_sector_rdms -- the E_pq psi vectors are formed
 *                          again, chunk by chunk in LDS, never in memory)
 * work: oovqe_sector_work_size() doubles. */
int oovqe_sector_state(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                       int ncas, uint32_t init_index, const uint32_t* unrank_a, const uint32_t* unrank_b,
                       const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                       double* psi_c, double* psi_dense, oovqe_stream_t stream);
/* The sector circuit with up to two of its gates differentiated (first / second tangent states d psi / d theta_j,
 * d^2 psi / d theta_j d theta_k for the second derivatives of src/auto_oo/oo_pqc.py:103-125; circuits in which
 * every parameter drives one gate: UCCD / UCCSD / kUpCCD): deriv [n_out][2] gate indices (device; -1 = none,
 * twice the same gate = its second derivative), psi_out [batch][n_out][Dc]. */
int oovqe_sector_state_deriv(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                             int ncas, uint32_t init_index, const uint32_t* unrank_a,
                             const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                             int na, int nb, int batch, const int32_t* deriv, int n_out, double* psi_out,
                             oovqe_stream_t stream);
int oovqe_sector_rdms(const double* psi_c, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                      const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                      double* gamma, double* Gamma, double* work, oovqe_stream_t stream);
int oovqe_sector_adjoint(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                         int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                         const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                         const double* psi_c, const double* c1, const double* c2, double* work,
                         double* dtheta, oovqe_stream_t stream);
int64_t oovqe_sector_work_size(int ncas, int na, int nb, int batch);
/* Round 4: the gates of a circuit as PAIR LISTS.  A Givens pass rotates the pairs (d, e) of determinants whose
 * occupations match the gate's masks -- 400 of the 4 900 determinants of CAS(8e,8o) for a pair double excitation;
 * which ones depends on the gate table alone.  oovqe_sector_pairs lists them once per circuit
 * (pairs: oovqe_sector_pairs_size() 32-bit words = [n_gates] counts, then per gate its pairs d | e << 15 |
 * parity << 31 in ascending d; at most 32 767 determinants); oovqe_sector_state_pl / _state_deriv_pl / _adjoint_pl are
 * oovqe_sector_state / _state_deriv / _adjoint sweeping the lists (max_pairs = the largest count; pairs == NULL: the
 * plain entry points).  Same results to rounding (the sums of the adjoint run in another order). */
/* lam [batch][Dc] = (Hop + Hop^T) v for a stack of sector vectors v, Hop = sum c1e_pq E_pq + sum c2_pqrs E_pq E_rs the
 * operator whose quadratic form is Q(v) = c1 . gamma(v) + c2 . Gamma(v) (oo_pqc.py:103-111 differentiates exactly
 * that): the first stage of oovqe_sector_adjoint on its own.  a^T lam(b) is the symmetric bilinear form second
 * derivatives are made of: d^2 Q / dtheta_j dtheta_k = tau_jk^T lam(psi) + tau_j^T lam(tau_k). */
int oovqe_sector_lambda(const double* vecs, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                        const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch, const double* c1,
                        const double* c2, const uint16_t* tabs, double* work, double* lam, oovqe_stream_t stream);
/* The same two for a STACK OF GEOMETRIES, each with its own CAS coefficients (the sector engine under the
 * geometry batch: ansatze/kUpCCD.py:36-154 under oo_pqc.py:64-148, one launch sequence for all geometries): state b
 * takes c1 + (b / group) * c_stride and c2 + (b / group) * c_stride (doubles; e.g. columns of the packed outputs of
 * oovqe_cas_eval_batch).  oovqe_sector_adjoint_pg: group = 1.  oovqe_sector_lambda_pg with group = 1 + n_theta: the
 * operator on psi and its first tangents of every geometry (theta-theta blocks).  Served for ncas = 4 and 8
 * (oovqe_sector_geometry_coefficients_ok); callers loop over the geometries otherwise. */
int oovqe_sector_geometry_coefficients_ok(int ncas, int na, int nb);
int oovqe_sector_adjoint_pg(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                            const uint32_t* unrank_a, const uint32_t* unrank_b, const int32_t* rank_a,
                            const int32_t* rank_b, int na, int nb, int batch, const double* psi_c, const double* c1,
                            const double* c2, int64_t c_stride, const uint32_t* pairs, int max_pairs,
                            const uint16_t* tabs, double* work, double* dtheta, oovqe_stream_t stream);
int oovqe_sector_lambda_pg(const double* vecs, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                           const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch, int group,
                           const double* c1, const double* c2, int64_t c_stride, const uint16_t* tabs, double* work,
                           double* lam, oovqe_stream_t stream);
int64_t oovqe_sector_pairs_size(int n_gates, int na, int nb);
/* oovqe_sector_pairs also leaves the sector's two excitation tables ([a^2][na] | [a^2][nb], 16-bit: source string,
 * valid bit, parities) behind the lists -- its buffer must hold oovqe_sector_pairs_size + oovqe_sector_tables_size
 * words, the tables start at word oovqe_sector_pairs_size.  Passed as `tabs` (NULL: every workgroup builds them from
 * the strings) to oovqe_sector_rdms_tb / oovqe_sector_adjoint_pl / oovqe_sector_lambda. */
int64_t oovqe_sector_tables_size(int ncas, int na, int nb);
int oovqe_sector_rdms_tb(const double* psi_c, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                         const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch, const uint16_t* tabs,
                         double* gamma, double* Gamma, double* work, oovqe_stream_t stream);
int oovqe_sector_pairs(const oovqe_gate_t* gates, int n_gates, int ncas, const uint32_t* unrank_a,
                       const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b, int na, int nb,
                       uint32_t* pairs, oovqe_stream_t stream);
int oovqe_sector_state_pl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                          uint32_t init_index, const uint32_t* unrank_a, const uint32_t* unrank_b,
                          const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                          const uint32_t* pairs, int max_pairs, double* psi_c, double* psi_dense,
                          oovqe_stream_t stream);
int oovqe_sector_state_deriv_pl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                                uint32_t init_index, const uint32_t* unrank_a, const uint32_t* unrank_b,
                                const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                                const uint32_t* pairs, int max_pairs, const int32_t* deriv, int n_out,
                                double* psi_out, oovqe_stream_t stream);
int oovqe_sector_adjoint_pl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                            const uint32_t* unrank_a, const uint32_t* unrank_b, const int32_t* rank_a,
                            const int32_t* rank_b, int na, int nb, int batch, const double* psi_c,
                            const double* c1, const double* c2, const uint32_t* pairs, int max_pairs,
                            const uint16_t* tabs, double* work, double* dtheta, oovqe_stream_t stream);

/* ---- a12/a13/a14/a16: one evaluation of the hybrid cost function in ONE call ---------------------
 * OO_pqc.energy_from_parameters / circuit_gradient / orbital_gradient / orbital_circuit_hessian
 * (src/auto_oo/oo_pqc.py:64-125) for a single geometry: circuit (+ tangents when derivatives != 0)
 * -> RDM sets -> CAS path.  5 launches on `stream`, no host synchronisation.
 * out (packed): [c0 | E | dE/dtheta (n_theta, or 1 slot when derivatives == 0) |
 *                gvec (nvec x n_kappa; row 0 = dE/dkappa, rows k>=1 = d^2E/dkappa dtheta_k) |
 *                c1 (a^2) | c2 (a^4)],  nvec = derivatives ? 1 + n_theta : 1.
 * work: oovqe_oo_eval_work_size() doubles. */
int oovqe_oo_eval(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                  int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                  const double* C, double nuc, int N, int n_occ, int ncas, const int32_t* kap_row,
                  const int32_t* kap_col, int n_kappa, int derivatives, double* work, double* out,
                  unsigned eri_flags, oovqe_stream_t stream);
int64_t oovqe_oo_eval_work_size(int n_theta, int n_gates, int n_qubits, int N, int n_occ, int ncas,
                                int derivatives);
/* The CAS path of a STACK of geometries from GIVEN RDM sets (circuits whose state lives in the sector engine:
 * kUpCCD CAS(8e,8o) over the geometries of a Berry-phase loop): gamma [G][nrdm][a^2], Gamma [G][nrdm][a^4] (set 0 =
 * the RDMs, sets k >= 1 = derivative RDMs), nuc [G] (device); out [G][oovqe_oo_eval_out_size(nrdm - 1, n_kappa, ncas,
 * nrdm > 1)] in the packed layout of oovqe_oo_eval_batch; work G * oovqe_cas_eval_work_size(N, n_occ, ncas, nrdm);
 * fock [G][N][N] or NULL.  Replaces, per geometry, what oovqe_cas_eval replaces (oo_energy.py:178-309). */
int oovqe_cas_eval_batch(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                         const double* Gamma, int nrdm, const double* nuc, int N, int n_occ, int ncas,
                         const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int batch, double* work,
                         double* out, double* fock, unsigned eri_flags, const double* g_packed,
                         oovqe_stream_t stream);
int64_t oovqe_oo_eval_out_size(int n_theta, int n_kappa, int ncas, int derivatives);
/* The same for a BATCH of geometries in one call (the Berry-phase-loop batch of the north star;
 * examples/Tutorial_Berry_phase.ipynb): every per-geometry array is stacked along a leading batch
 * axis -- theta [G,n_theta], g_ao [G,N^4], h_ao [G,N^2], C [G,N^2], nuc [G] (device), work
 * G * oovqe_oo_eval_work_size(), out [G, oovqe_oo_eval_out_size()]; g_packed [G, oovqe_eri_packed_size()]
 * or NULL.  Still 5 launches: the batch is a
 * grid dimension of every kernel, so small geometries fill the 256 CUs together. */
int oovqe_oo_eval_batch(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                        int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                        const double* C, const double* nuc, int N, int n_occ, int ncas,
                        const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int derivatives,
                        int batch, double* work, double* out, unsigned eri_flags,
                        const double* g_packed, oovqe_stream_t stream);
/* ---- configs[3]'s unit of work for a batch of geometries: energy + full gradient + full Hessian ----
 * OO_pqc.full_gradient + OO_pqc.full_hessian (src/auto_oo/oo_pqc.py:132-148) of every geometry of a
 * stack in ONE call: the batched evaluation above with derivatives (E, dE/dtheta, dE/dkappa, the
 * orbital-circuit block, c1, c2, generalized Fock matrices), the circuit-circuit block from second
 * tangent states (oo_pqc.py:103-111) and the orbital-orbital block (oo_energy.py:311-402) -- every
 * launch carries the geometry index as a grid dimension.  Arguments as oovqe_oo_eval_batch; pairs
 * [n_pairs][2] = every (j <= k) of the theta indices; out [batch][oovqe_oo_eval_out_size(.., 1)] (the
 * packed evaluation result: [E | full gradient] is out[b][1 : 2 + n_theta + n_kappa]); hessian
 * [batch][n][n], n = n_theta + n_kappa, laid out [[theta-theta, (kappa-theta)^T], [kappa-theta,
 * kappa-kappa]] as the reference's full_hessian; work: batch * oovqe_oo_hessian_work_size(). */
int oovqe_oo_hessian_batch(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                           int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                           const double* C, const double* nuc, int N, int n_occ, int ncas,
                           const int32_t* kap_row, const int32_t* kap_col, int n_kappa,
                           const int32_t* pairs, int n_pairs, int batch, double* work, double* out,
                           double* hessian, unsigned eri_flags, const double* g_packed,
                           oovqe_stream_t stream);
int64_t oovqe_oo_hessian_work_size(int n_theta, int n_gates, int n_qubits, int N, int n_occ, int ncas,
                                   int n_pairs);
/* Orbital rotation of a stack of geometries in one launch (OO_energy.get_transformed_mo,
 * src/auto_oo/oo_energy.py:213-236, per geometry): C_out[b] = C[b] expm(-K(kappa[b])); kappa
 * [batch][n_kappa], C / C_out [batch][N][N] (C_out may alias C when N <= 48), U [batch][N][N] or NULL
 * (the rotation matrices).  work: NULL for N <= 48 (one workgroup per geometry, all in LDS), else
 * (batch + 7) * N * N doubles. */
int oovqe_rotate_orbitals_batch(const double* kappa, const int32_t* kap_row, const int32_t* kap_col,
                                int n_kappa, int N, int batch, const double* C, double* C_out, double* U,
                                double* work, oovqe_stream_t stream);
/* ---- (e)/(f1): one damped Newton step of a stack of geometries in lockstep, enqueued by ONE call -------------
 * The body of OO_pqc.full_optimization / of the Berry-phase loop (src/auto_oo/oo_pqc.py:172-196,
 * examples/Tutorial_Berry_phase.ipynb raw 408-441) per geometry, with NewtonStep.damped_newton_step
 * (src/auto_oo/utils/newton_raphson.py:194-211) up to the FIRST verdict of its line search: E + full gradient + full
 * Hessian (oovqe_oo_hessian_batch), the directions (Cholesky fast path, band route for the others), the trial
 * points flat + dp with their orbitals C_oao expm(-K), S^-1/2 (C_oao U) (oo_pqc.py:191, oo_energy.py:173-176), the
 * trial energies and the acceptance rule (newton_raphson.py:146-177) -- the launches of the single entry points,
 * back to back on `stream` with no host work between them (at the 8 geometries per rank of an 8-GPU job the host's
 * share of a step driven call by call was a fifth of it).  The host then reads flags [4] (oovqe_linesearch_update):
 * flags[0] == 0: every problem accepted its first trial -- new parameters = points_a / points_b, new orbitals =
 * trial_oao / trial_mo, new energies = state[batch .. 2 batch).  Otherwise it continues with the single entry points.
 * side_stream (may be NULL): the lowest eigenvalues of positive definite Hessians -- a reported number no step reads
 * (newton_raphson.py:105-128) -- are computed there with at most side_wg workgroups per problem (0: no bound).
 * speculate != 0: the band route of the problems the fast path did NOT serve runs on the side stream too, and the
 * trial is formed without waiting for it: when flags[1] (min of info) comes back <= 0 the caller must join the side
 * stream and repeat the trial (for loops whose Hessians stay positive definite: nothing but the fast path is then
 * on the calling stream).
 * All arrays on the device, stacked over the batch; sizes from the *_work_size / *_out_size functions of the entry
 * points named above. */
typedef struct oovqe_newton_step_t {
    /* circuit and integrals, as oovqe_oo_hessian_batch */
    const double* theta;            /* [batch][n_theta] */
    const oovqe_gate_t* gates;
    const double* g_ao;
    const double* h_ao;
    const double* nuc;              /* [batch] */
    const double* g_packed;         /* or NULL */
    /* orbitals: S^-1/2 [batch][N][N], C_oao [batch][N][N], mo_coeff = S^-1/2 C_oao [batch][N][N] */
    const double* oao_coeff;
    const double* oao_mo_coeff;
    const double* mo_coeff;
    const int32_t* kap_row;
    const int32_t* kap_col;
    const int32_t* pairs;           /* [n_pairs][2], every j <= k */
    /* workspaces */
    double* work_hessian;           /* batch * oovqe_oo_hessian_work_size() */
    double* work_eval;              /* batch * oovqe_oo_eval_work_size(derivatives = 0) */
    double* work_pd;                /* oovqe_newton_direction_pd_work_size(n, batch), or NULL: band route only */
    double* work_rest;              /* oovqe_newton_direction_rest_work_size(n, batch) */
    double* work_rest_side;         /* the same size, for the side stream (NULL without one) */
    double* work_rotate;            /* NULL for N <= 48, else (batch + 7) N^2 */
    /* results */
    double* out;                    /* [batch][oovqe_oo_eval_out_size(derivatives = 1)] */
    double* hessian;                /* [batch][n][n], n = n_theta + n_kappa */
    double* grad;                   /* [batch][n] */
    double* energy;                 /* [batch] */
    double* flat;                   /* [batch][n] = [theta | 0] */
    double* dp;                     /* [batch][n] */
    double* lowest;                 /* [batch] (side stream) */
    double* shift;                  /* [batch] */
    double* info;                   /* [batch], as oovqe_newton_direction_rest */
    double* t;                      /* [batch] step lengths */
    double* state;                  /* [3][batch]: still searching | energy at the accepted point | Armijo slope */
    double* flags;                  /* [4], as oovqe_linesearch_update */
    double* points_a;               /* [batch][n_theta] */
    double* points_b;               /* [batch][n_kappa] */
    double* trial_oao;              /* [batch][N][N] */
    double* trial_mo;               /* [batch][N][N] */
    double* trial_out;              /* [batch][oovqe_oo_eval_out_size(derivatives = 0)] */
    /* reference defaults: newton_raphson.py:47-61 */
    double lambda_min, mu, rho, alpha, beta;
    int32_t n_theta, n_gates, n_qubits, N, n_occ, ncas, n_kappa, n_pairs, batch, aug, speculate, side_wg;
    uint32_t init_index, eri_flags;
} oovqe_newton_step_t;
int oovqe_oo_newton_step_batch(const oovqe_newton_step_t* step, oovqe_stream_t stream,
                               oovqe_stream_t side_stream);
/* sizeof(oovqe_newton_step_t) in the library as built (a binding compares its mirror of the block with it) */
int oovqe_newton_step_size(void);

/* 1 when oovqe_circuit_rdms takes its one-workgroup LDS path for these sizes */
int oovqe_circuit_rdms_is_small(int n_qubits, int ncas, int nvec, int n_gates);

#ifdef __cplusplus
}
#endif
#endif /* OOVQE_H */
