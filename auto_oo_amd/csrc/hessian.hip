// Orbital-orbital Hessian.
//
// Reference: OO_energy.full_rdms / y_matrix / analytic_hessian_from_integrals /
// full_hessian_to_matrix (src/auto_oo/oo_energy.py:311-402):
//   H_pqrs = (1-P_pq)(1-P_rs) [ 2 gam_pr h_qs - (F_pr + F_rp) d_qs + 2 Y_pqrs ]
//   Y_pqrs = sum_mn [ (Gam_pmrn + Gam_pmnr) g_qmns + Gam_prmn g_qsmn ]
// with the full-space RDMs of oo_energy.py:342-379.  The reference evaluates Y with three dense
// N^6 einsums over N^4 tensors (38 GFLOP at N = 43).  The full-space 2-RDM is supported on
// (occ+act)^4 only, so Y_pqrs vanishes unless p, r < M = n_occ + ncas and needs only MO integrals
// with TWO general indices:
//   Jint[q,s,m,n] = g_mo[q,s,m,n]      Kint[q,m,n,s] = g_mo[q,m,n,s]        (m, n < M)
// Both are produced from g_ao with partial transforms on the K1 contraction kernel
// (O(N^4 M) flop), Y is two small GEMMs (O(N^2 M^4)), and an assembly kernel applies the
// antisymmetrisers directly on the non-redundant (tril) pairs.
#include "common.h"

int oovqe_mode_contract_impl(const double* T, const double* Cm, double* out, long A, int K, int J,
                             long B, int ldc, int last, hipStream_t st);

namespace {

// full-space 2-RDM on (occ+act)^4, oo_energy.py:363-378 (all other elements are zero)
__device__ __forceinline__ double gamma2_full(int p, int m, int r, int n, int no, int na,
                                              const double* __restrict__ gam,
                                              const double* __restrict__ Gam)
{
    const bool po = p < no, mo = m < no, ro = r < no, nn = n < no;
    if (po && mo && ro && nn) return 4.0 * (p == m && r == n) - 2.0 * (p == n && m == r);
    if (po && mo && !ro && !nn) return (p == m) ? 2.0 * gam[(r - no) * na + (n - no)] : 0.0;
    if (!po && !mo && ro && nn) return (r == n) ? 2.0 * gam[(p - no) * na + (m - no)] : 0.0;
    if (po && !mo && !ro && nn) return (p == n) ? -gam[(m - no) * na + (r - no)] : 0.0;
    if (!po && mo && ro && !nn) return (m == r) ? -gam[(n - no) * na + (p - no)] : 0.0;
    if (!po && !mo && !ro && !nn)
        return Gam[(((p - no) * na + (m - no)) * na + (r - no)) * na + (n - no)];
    return 0.0;
}

// At[(m,n),(p,r)] = Gam_pmrn + Gam_pmnr ;  Bt[(m,n),(p,r)] = Gam_prmn      (all indices < M)
__global__ void hess_ab_kernel(const double* __restrict__ gam, const double* __restrict__ Gam,
                               int no, int na, double* __restrict__ At, double* __restrict__ Bt)
{
    const int M = no + na;
    const long total = (long)M * M * M * M;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int r = (int)(t % M); t /= M;
        const int p = (int)(t % M); t /= M;
        const int n = (int)(t % M); t /= M;
        const int m = (int)t;
        At[idx] = gamma2_full(p, m, r, n, no, na, gam, Gam) + gamma2_full(p, m, n, r, no, na, gam, Gam);
        Bt[idx] = gamma2_full(p, r, m, n, no, na, gam, Gam);
    }
}

struct HessArgs {
    const double* YkT;    // [q][(p,r)][s]   = sum_mn At[(mn),(pr)] Kint[q,m,n,s]
    const double* YjT;    // [q][s][(p,r)]   = sum_mn Bt[(mn),(pr)] Jint[q,s,m,n]
    const double* hmo;    // [N][N]
    const double* fock;   // [N][N] generalized Fock (rows >= M are zero)
    const double* gam;    // [na][na]
    int N, no, na;
};

__device__ __forceinline__ double hess_x(const HessArgs& a, int p, int q, int r, int s)
{
    // X_pqrs = 2 gam_pr h_qs - (F_pr + F_rp) d_qs + 2 Y_pqrs
    const int M = a.no + a.na, N = a.N;
    double x = 0.0;
    if (q == s) x -= a.fock[(size_t)p * N + r] + a.fock[(size_t)r * N + p];
    if (p < M && r < M) {
        double g1 = 0.0;
        if (p < a.no && r < a.no) g1 = (p == r) ? 2.0 : 0.0;
        else if (p >= a.no && r >= a.no) g1 = a.gam[(p - a.no) * a.na + (r - a.no)];
        x += 2.0 * g1 * a.hmo[(size_t)q * N + s];
        const size_t pr = (size_t)p * M + r;
        x += 2.0 * (a.YkT[((size_t)q * M * M + pr) * N + s] + a.YjT[((size_t)q * N + s) * M * M + pr]);
    }
    return x;
}

// H[t1,t2] on the non-redundant (row>col) pairs
__global__ void hess_matrix_kernel(HessArgs a, const int32_t* __restrict__ kap_row,
                                   const int32_t* __restrict__ kap_col, int n_kappa,
                                   double* __restrict__ H)
{
    const long total = (long)n_kappa * n_kappa;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int t1 = (int)(idx / n_kappa), t2 = (int)(idx - (long)t1 * n_kappa);
        const int p = kap_row[t1], q = kap_col[t1], r = kap_row[t2], s = kap_col[t2];
        H[idx] = hess_x(a, p, q, r, s) - hess_x(a, p, q, s, r) - hess_x(a, q, p, r, s) +
                 hess_x(a, q, p, s, r);
    }
}

// full [N,N,N,N] tensor (what OO_energy.analytic_hessian returns, oo_energy.py:335-340)
__global__ void hess_full_kernel(HessArgs a, double* __restrict__ H)
{
    const int N = a.N;
    const long total = (long)N * N * N * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int s = (int)(t % N); t /= N;
        const int r = (int)(t % N); t /= N;
        const int q = (int)(t % N); t /= N;
        const int p = (int)t;
        H[idx] = hess_x(a, p, q, r, s) - hess_x(a, p, q, s, r) - hess_x(a, q, p, r, s) +
                 hess_x(a, q, p, s, r);
    }
}

}  // namespace

extern "C" int64_t oovqe_orbital_hessian_work_size(int N, int n_occ, int ncas)
{
    const int64_t M = n_occ + ncas, n = N;
    // T2, Uj, Jint (n^2 M^2 each) | Vk (n^3 M) | T2K, W, Kint (n^2 M^2 each) | X1, hmo (n^2 each)
    // | At, Bt (M^4 each) | YkT, YjT (n^2 M^2 each)
    return 8 * n * n * M * M + n * n * n * M + 2 * n * n + 2 * M * M * M * M;
}

extern "C" int oovqe_cas_half_transform(const double* g_ao, const double* C, int N, int M, double* T2,
                                        oovqe_stream_t stream);

extern "C" int oovqe_orbital_hessian(const double* g_ao, const double* h_ao, const double* C,
                                     const double* gamma, const double* Gamma, const double* fock,
                                     int N, int n_occ, int ncas, const int32_t* kap_row,
                                     const int32_t* kap_col, int n_kappa, double* work,
                                     double* H_matrix, double* H_full, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && h_ao && C && gamma && Gamma && fock && work, "orbital_hessian: null pointer");
    OOVQE_REQUIRE(H_matrix || H_full, "orbital_hessian: no output requested");
    OOVQE_REQUIRE(!H_matrix || (kap_row && kap_col && n_kappa > 0), "orbital_hessian: index tables");
    OOVQE_REQUIRE(N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N, "orbital_hessian: sizes");
    hipStream_t st = (hipStream_t)stream;
    const int M = n_occ + ncas;
    const long n = N, m2 = (long)M * M, n2 = n * n;
    double* T2 = work;                 // [N][N][M][M]
    double* Uj = T2 + n2 * m2;         // [N][N][M*M]
    double* Jint = Uj + n2 * m2;       // [N][N][M*M]      g_mo[q,s,m,n]
    double* Vk = Jint + n2 * m2;       // [N][N][M][N]
    double* T2K = Vk + n2 * n * M;     // [N][M][M][N]
    double* W = T2K + n2 * m2;         // [N][M][M][N]
    double* Kint = W + n2 * m2;        // [N][M][M][N]     g_mo[q,m,n,s]
    double* X1 = Kint + n2 * m2;       // [N][N]
    double* hmo = X1 + n2;             // [N][N]
    double* At = hmo + n2;             // [M*M][M*M]
    double* Bt = At + m2 * m2;         // [M*M][M*M]
    double* YkT = Bt + m2 * m2;        // [N][M*M][N]
    double* YjT = YkT + n2 * m2;       // [N][N][M*M]
    int rc;
#define MC(...) if ((rc = oovqe_mode_contract_impl(__VA_ARGS__, st))) return rc
    // ---- J-type integrals: g_mo[q,s,m,n] ---------------------------------------------------------
    if ((rc = oovqe_cas_half_transform(g_ao, C, N, M, T2, stream))) return rc;   // T2[p,q,y,z]
    MC(T2, C, Uj, 1, N, N, n * m2, N, 0);          // Uj[q',q,yz]  = sum_p C[p,q'] T2[p,q,yz]
    MC(Uj, C, Jint, n, N, N, m2, N, 0);            // Jint[q',s',yz] = sum_q C[q,s'] Uj[q',q,yz]
    // ---- K-type integrals: g_mo[q,m,n,s] ---------------------------------------------------------
    MC(g_ao, C, Vk, n2, N, M, n, N, 0);            // Vk[p,q,n,s]  = sum_r C[r,n] g[p,q,r,s]
    MC(Vk, C, T2K, n, N, M, (long)M * n, N, 0);    // T2K[p,m,n,s] = sum_q C[q,m] Vk[p,q,n,s]
    MC(T2K, C, W, n * m2, N, N, 1, N, 1);          // W[p,m,n,s']  = sum_s T2K[p,m,n,s] C[s,s']
    MC(W, C, Kint, 1, N, N, m2 * n, N, 0);         // Kint[q',m,n,s'] = sum_p C[p,q'] W[p,m,n,s']
    // ---- one-electron integrals ------------------------------------------------------------------
    MC(h_ao, C, X1, 1, N, N, n, N, 0);             // X1 = C^T h
    MC(X1, C, hmo, n, N, N, 1, N, 1);              // hmo = X1 C
    // ---- Y ------------------------------------------------------------------------------------------
    hess_ab_kernel<<<64, 256, 0, st>>>(gamma, Gamma, n_occ, ncas, At, Bt);
    // YkT[q,(pr),s] = sum_(mn) At[(mn),(pr)] Kint[q,(mn),s]
    MC(Kint, At, YkT, n, (int)m2, (int)m2, n, (int)m2, 0);
    // YjT[(qs),(pr)] = sum_(mn) Jint[(qs),(mn)] Bt[(mn),(pr)]
    MC(Jint, Bt, YjT, n2, (int)m2, (int)m2, 1, (int)m2, 1);
#undef MC
    HessArgs a{YkT, YjT, hmo, fock, gamma, N, n_occ, ncas};
    if (H_matrix) {
        const long total = (long)n_kappa * n_kappa;
        const unsigned nb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hess_matrix_kernel<<<nb, 256, 0, st>>>(a, kap_row, kap_col, n_kappa, H_matrix);
    }
    if (H_full) {
        const long total = n2 * n2;
        const unsigned nb = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hess_full_kernel<<<nb, 256, 0, st>>>(a, H_full);
    }
    OOVQE_CHECK_LAUNCH("orbital_hessian");
    return 0;
}
