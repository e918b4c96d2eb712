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
// blockIdx.y = geometry of a batch (gam / Gam / At / Bt advance by their batch strides)
__global__ void hess_ab_kernel(const double* __restrict__ gam, const double* __restrict__ Gam,
                               int no, int na, double* __restrict__ At, double* __restrict__ Bt,
                               long gam_bs, long Gam_bs)
{
    const int M = no + na;
    const long total = (long)M * M * M * M;
    gam += (size_t)blockIdx.y * gam_bs;
    Gam += (size_t)blockIdx.y * Gam_bs;
    At += (size_t)blockIdx.y * total;
    Bt += (size_t)blockIdx.y * total;
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
    // batch strides (doubles) of the five arrays above; geometry = blockIdx.y
    long y_bs, h_bs, f_bs, g_bs;
};

__device__ __forceinline__ HessArgs hess_batch_slice(HessArgs a, unsigned b)
{
    a.YkT += (size_t)b * a.y_bs;
    a.YjT += (size_t)b * a.y_bs;
    a.hmo += (size_t)b * a.h_bs;
    a.fock += (size_t)b * a.f_bs;
    a.gam += (size_t)b * a.g_bs;
    return a;
}

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

// H[t1,t2] on the non-redundant (row>col) pairs; element (t1,t2) of geometry b goes to
// H[b * h_out_bs + t1 * ldh + t2] (a block of a larger matrix when ldh > n_kappa)
__global__ void hess_matrix_kernel(HessArgs a0, const int32_t* __restrict__ kap_row,
                                   const int32_t* __restrict__ kap_col, int n_kappa,
                                   double* __restrict__ H, long ldh, long h_out_bs)
{
    const HessArgs a = hess_batch_slice(a0, blockIdx.y);
    H += (size_t)blockIdx.y * h_out_bs;
    const long total = (long)n_kappa * n_kappa;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int t1 = (int)(idx / n_kappa), t2 = (int)(idx - (long)t1 * n_kappa);
        const int p = kap_row[t1], q = kap_col[t1], r = kap_row[t2], s = kap_col[t2];
        H[(size_t)t1 * ldh + t2] = hess_x(a, p, q, r, s) - hess_x(a, p, q, s, r) - hess_x(a, q, p, r, s) +
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

// kappa-theta block of the full Hessian from the packed evaluation output (gvec rows k >= 1 =
// d G_kappa / d theta_k, oo_pqc.py:113-125,141-148): H[nt + i][k] = H[k][nt + i] = gvec[1 + k][i]
__global__ void hess_cross_block_kernel(const double* __restrict__ gvec, long gvec_bs, int n_theta,
                                        int n_kappa, double* __restrict__ H, long ldh, long h_bs)
{
    gvec += (size_t)blockIdx.y * gvec_bs;
    H += (size_t)blockIdx.y * h_bs;
    const long total = (long)n_theta * n_kappa;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx / n_kappa), i = (int)(idx - (long)k * n_kappa);
        const double v = gvec[(size_t)(1 + k) * n_kappa + i];
        H[(size_t)(n_theta + i) * ldh + k] = v;
        H[(size_t)k * ldh + n_theta + i] = v;
    }
}

// T2K[p,m,n,s] = sum_q C[q,m] Vk[p,q,n,s] from the quarter-transformed slabs p <= q that stage 1 leaves
// behind (cas.hip, half_transform_kernel: Vk_tri[tri(p,q)][s][n]; g[p,q,:,:] == g[q,p,:,:]): one workgroup
// per (p, geometry), thread <-> element (s, n) of a row of 43 slabs, the coefficients wave-uniform; the
// block T2K[p] (contiguous) leaves through LDS.  Replaces two K1 launches that read the whole AO tensor a
// second time (round 3: 480 + 102 us -> see DESIGN.md for 64 geometries).
template <int MT>                              // accumulators per thread: the smallest of 4, 8, 12, 16 >= M
__global__ __launch_bounds__(512)
void t2k_tri_kernel(const double* __restrict__ Vk, const double* __restrict__ C, double* __restrict__ T2K,
                    int N, int M)
{
    extern __shared__ double blk[];            // [M][M][N]
    const int p = blockIdx.x, tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const long tri = (long)N * (N + 1) / 2;
    const int E = N * M;
    Vk += b * (size_t)tri * E;
    C += b * (size_t)N * N;
    T2K += (b * N + p) * (size_t)M * M * N;
    for (int e = tid; e < E; e += 512) {
        const int s = e / M, y = e - s * M;
        double acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.0;
        // sixteen rows in flight (8: 150 us, 16: 108, 24: 111 for 64 geometries); straight-line sums (columns m >= M of C are other coefficients: summed,
        // never stored; rows q >= N enter with weight zero)
        for (int q0 = 0; q0 < N; q0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = q0 + u < N ? q0 + u : N - 1;
                const int lo = p < q ? p : q, hi = p < q ? q : p;
                const long t = (long)lo * (2 * N - lo + 1) / 2 + (hi - lo);
                v[u] = Vk[(size_t)t * E + e];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = q0 + u < N ? q0 + u : N - 1;
                const double vv = q0 + u < N ? v[u] : 0.0;
                const double* cq = C + (size_t)q * N;              // wave-uniform: scalar loads
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] += cq[m < N ? m : 0] * vv;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < M) blk[((size_t)m * M + y) * N + s] = acc[m];
    }
    __syncthreads();
    for (int idx = tid; idx < M * M * N; idx += 512) T2K[idx] = blk[idx];
}

}  // namespace

extern "C" int64_t oovqe_orbital_hessian_work_size(int N, int n_occ, int ncas)
{
    const int64_t M = n_occ + ncas, n = N;
    // T2, Uj, Jint (n^2 M^2 each) | Vk (n^3 M) | T2K, W, Kint (n^2 M^2 each) | X1, hmo (n^2 each)
    // | At, Bt (M^4 each) | YkT, YjT (n^2 M^2 each)
    return 8 * n * n * M * M + n * n * n * M + 2 * n * n + 2 * M * M * M * M;
}

// cas.hip: stage 1 for a stack of geometries (slabs p <= q only when the flags allow it)
int oovqe_half_transform_batched_impl(const double* g_ao, const double* C, int N, int M, double* T2,
                                      int batch, unsigned eri_flags, oovqe_stream_t stream,
                                      double* Vk_tri = nullptr);
int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);

// The orbital-orbital Hessian for `batch` geometries of identical shape: every input stacked along a
// leading batch axis with the given strides (doubles), every intermediate stacked in `work`
// (batch * oovqe_orbital_hessian_work_size() doubles); the geometry index is a grid dimension of
// every launch.  H_matrix element (t1,t2) of geometry b -> H_matrix[b * h_bs + t1 * ldh + t2].
// The chains of stage 2 that do not depend on each other, beside each other (oovqe_oo_hessian_batch): the K-type AND
// J-type integrals, the Y products and the assembly on sK (ONE internal stream of the library: HIP multiplexes a
// process' streams over four hardware queues, and the caller's stream + the two side streams of the eigenvalue route
// leave one), the one-electron integrals on the caller's stream behind the evaluation.  sK must already wait for
// stage 1 (T2 / Vk); its only waits are for the RDMs (recorded inside the evaluation, behind the launch the circuit
// rides on) and, before the last kernel, for fock + h_mo; done: recorded on sK behind the last launch.
struct HessFork {
    hipStream_t sK;
    hipEvent_t rdm_ready, rest_ready, done;      // RDMs complete | fock and h_mo complete (caller's stream) | block written (sK)
};

static int orbital_hessian_batched(const double* g_ao, const double* h_ao, const double* C,
                                   const double* gamma, long gamma_bs, const double* Gamma, long Gamma_bs,
                                   const double* fock, int N, int n_occ, int ncas, const int32_t* kap_row,
                                   const int32_t* kap_col, int n_kappa, int batch, double* work,
                                   double* H_matrix, long ldh, long h_bs, double* H_full,
                                   unsigned eri_flags, oovqe_stream_t stream, int stage = 0,
                                   HessFork* fork = nullptr)
{
    // stage 0: everything; 1: stage 1 of the integrals only (T2 and, when available, Vk: needs neither
    // the RDMs nor the Fock matrices); 2: the rest, on the T2 / Vk a stage-1 call left in `work`; 3 (with a fork):
    // the K-type chain alone, on the fork's stream (needs stage 1 only: the caller enqueues it FIRST -- the host hands
    // out launches at a few microseconds apiece, and what sits at the back of the host's order starts late whatever
    // the streams allow); a stage-2 call with a fork then skips that chain
    OOVQE_REQUIRE(g_ao && h_ao && C && work && (stage == 1 || stage == 3 || (gamma && Gamma && fock)),
                  "orbital_hessian: null pointer");
    OOVQE_REQUIRE(stage == 1 || stage == 3 || H_matrix || H_full, "orbital_hessian: no output requested");
    OOVQE_REQUIRE(stage != 3 || fork, "orbital_hessian: stage 3 needs a fork");
    OOVQE_REQUIRE(!H_matrix || (kap_row && kap_col && n_kappa > 0), "orbital_hessian: index tables");
    OOVQE_REQUIRE(N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N, "orbital_hessian: sizes");
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535 && (batch == 1 || !H_full), "orbital_hessian: batch=%d", batch);
    hipStream_t st = (hipStream_t)stream;
    const int M = n_occ + ncas;
    const long n = N, m2 = (long)M * M, n2 = n * n;
    const size_t nb = (size_t)batch;
    // every block stacked over the batch
    double* T2 = work;                      // [G][N][N][M][M]
    double* Uj = T2 + nb * n2 * m2;         // [G][N][N][M*M]
    double* Jint = Uj + nb * n2 * m2;       // [G][N][N][M*M]      g_mo[q,s,m,n]
    double* Vk = Jint + nb * n2 * m2;       // [G][N][N][M][N]
    double* T2K = Vk + nb * n2 * n * M;     // [G][N][M][M][N]
    double* W = T2K + nb * n2 * m2;         // [G][N][M][M][N]
    double* Kint = W + nb * n2 * m2;        // [G][N][M][M][N]     g_mo[q,m,n,s]
    double* X1 = Kint + nb * n2 * m2;       // [G][N][N]
    double* hmo = X1 + nb * n2;             // [G][N][N]
    double* At = hmo + nb * n2;             // [G][M*M][M*M]
    double* Bt = At + nb * m2 * m2;         // [G][M*M][M*M]
    double* YkT = Bt + nb * m2 * m2;        // [G][N][M*M][N]
    double* YjT = YkT + nb * n2 * m2;       // [G][N][N][M*M]
    int rc;
    const long t4 = n2 * n2, y = n2 * m2;
    // sk: the stream of the K-type and J-type chains, the Y products and the assembly (the caller's unless forked)
    hipStream_t sk = st;
    const bool forked = fork && stage >= 2;
    if (forked) {
        sk = fork->sK;
        OOVQE_REQUIRE(sk && fork->rdm_ready && fork->rest_ready && fork->done, "orbital_hessian: fork without stream / events");
    }
    const bool k_chain = !(fork && stage == 2), rest = stage != 3;
    hipStream_t ms = st;           // the stream the MC macro launches on
#define MC(T_, tb, C_, cb, O_, ob, ...) \
    if ((rc = oovqe_mode_contract_batched(T_, C_, O_, __VA_ARGS__, batch, tb, cb, ob, ms))) return rc
    // ---- J-type integrals: g_mo[q,s,m,n] ---------------------------------------------------------
    // p <-> q symmetric integrals, N <= 48: stage 1 also leaves its first products Vk[tri(p,q)][s][n]
    // (the quarter transform of the K-type path), so the AO tensor is read once instead of twice
    const bool vk_tri = (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) != 0 && N <= 48 && M <= 16 &&
                        (size_t)m2 * N * sizeof(double) <= 150 * 1024 && oovqe_opt(OOVQE_OPT_HESS_VK_PASS) == 0;
    if (stage < 2 &&
        (rc = oovqe_half_transform_batched_impl(g_ao, C, N, M, T2, batch, eri_flags, stream, vk_tri ? Vk : nullptr)))
        return rc;
    if (stage == 1) return 0;
    if (rest) {
    // ---- one-electron integrals (the caller's stream) ---------------------------------------------
    ms = st;
    MC(h_ao, n2, C, n2, X1, n2, 1, N, N, n, N, 0);           // X1 = C^T h
    MC(X1, n2, C, n2, hmo, n2, n, N, N, 1, N, 1);            // hmo = X1 C
    if (forked) OOVQE_CHECK_HIP(hipEventRecord(fork->rest_ready, st), "orbital_hessian: hipEventRecord");
    }
    ms = sk;
    // ---- K-type integrals: g_mo[q,m,n,s] ---------------------------------------------------------
    if (k_chain) {
    if (vk_tri) {
        const size_t lds_bytes = (size_t)m2 * N * sizeof(double);
#define OOVQE_LAUNCH_T2K(MT_)                                                                             \
    do {                                                                                                  \
        if (lds_bytes > 64 * 1024) {                                                                      \
            int rc_lds = oovqe_ensure_dynamic_lds((const void*)t2k_tri_kernel<MT_>, lds_bytes);           \
            if (rc_lds) return rc_lds;                                                                    \
        }                                                                                                 \
        t2k_tri_kernel<MT_><<<dim3(N, batch), 512, lds_bytes, sk>>>(Vk, C, T2K, N, M); /* T2K[p,m,n,s] */  \
    } while (0)
        if (M <= 4) OOVQE_LAUNCH_T2K(4);
        else if (M <= 8) OOVQE_LAUNCH_T2K(8);
        else if (M <= 12) OOVQE_LAUNCH_T2K(12);
        else OOVQE_LAUNCH_T2K(16);
#undef OOVQE_LAUNCH_T2K
        OOVQE_CHECK_LAUNCH("orbital_hessian/t2k");
    } else {
        MC(g_ao, t4, C, n2, Vk, n2 * n * M, n2, N, M, n, N, 0);  // Vk[p,q,n,s]  = sum_r C[r,n] g[p,q,r,s]
        MC(Vk, n2 * n * M, C, n2, T2K, y, n, N, M, (long)M * n, N, 0);   // T2K[p,m,n,s] = sum_q C[q,m] Vk[p,q,n,s]
    }
    MC(T2K, y, C, n2, W, y, n * m2, N, N, 1, N, 1);          // W[p,m,n,s']  = sum_s T2K[p,m,n,s] C[s,s']
    MC(W, y, C, n2, Kint, y, 1, N, N, m2 * n, N, 0);         // Kint[q',m,n,s'] = sum_p C[p,q'] W[p,m,n,s']
    }
    if (!rest) return 0;
    // ---- Y ------------------------------------------------------------------------------------------
    if (forked) OOVQE_CHECK_HIP(hipStreamWaitEvent(sk, fork->rdm_ready, 0), "orbital_hessian: hipStreamWaitEvent");
    hess_ab_kernel<<<dim3(64, batch), 256, 0, sk>>>(gamma, Gamma, n_occ, ncas, At, Bt, gamma_bs, Gamma_bs);
    // YkT[q,(pr),s] = sum_(mn) At[(mn),(pr)] Kint[q,(mn),s]
    MC(Kint, y, At, m2 * m2, YkT, y, n, (int)m2, (int)m2, n, (int)m2, 0);
    // ---- J-type integrals: g_mo[q,s,m,n] ---------------------------------------------------------
    MC(T2, y, C, n2, Uj, y, 1, N, N, n * m2, N, 0);          // Uj[q',q,yz]  = sum_p C[p,q'] T2[p,q,yz]
    MC(Uj, y, C, n2, Jint, y, n, N, N, m2, N, 0);            // Jint[q',s',yz] = sum_q C[q,s'] Uj[q',q,yz]
    // YjT[(qs),(pr)] = sum_(mn) Jint[(qs),(mn)] Bt[(mn),(pr)]
    MC(Jint, y, Bt, m2 * m2, YjT, y, n2, (int)m2, (int)m2, 1, (int)m2, 1);
    if (forked) OOVQE_CHECK_HIP(hipStreamWaitEvent(sk, fork->rest_ready, 0), "orbital_hessian: hipStreamWaitEvent");
#undef MC
    HessArgs a{YkT, YjT, hmo, fock, gamma, N, n_occ, ncas, y, n2, n2, gamma_bs};
    if (H_matrix) {
        const long total = (long)n_kappa * n_kappa;
        const unsigned nbk = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hess_matrix_kernel<<<dim3(nbk, batch), 256, 0, sk>>>(a, kap_row, kap_col, n_kappa, H_matrix, ldh, h_bs);
    }
    if (H_full) {
        const long total = n2 * n2;
        const unsigned nbk = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hess_full_kernel<<<nbk, 256, 0, sk>>>(a, H_full);
    }
    OOVQE_CHECK_LAUNCH("orbital_hessian");
    if (forked) OOVQE_CHECK_HIP(hipEventRecord(fork->done, sk), "orbital_hessian: hipEventRecord");
    return 0;
}

extern "C" int oovqe_orbital_hessian(const double* g_ao, const double* h_ao, const double* C,
                                     const double* gamma, const double* Gamma, const double* fock,
                                     int N, int n_occ, int ncas, const int32_t* kap_row,
                                     const int32_t* kap_col, int n_kappa, double* work,
                                     double* H_matrix, double* H_full, oovqe_stream_t stream)
{
    return orbital_hessian_batched(g_ao, h_ao, C, gamma, 0, Gamma, 0, fock, N, n_occ, ncas, kap_row, kap_col,
                                   n_kappa, 1, work, H_matrix, n_kappa, 0, H_full, 0, stream);
}

extern "C" int oovqe_orbital_hessian_batch(const double* g_ao, const double* h_ao, const double* C,
                                           const double* gamma, const double* Gamma, const double* fock,
                                           int N, int n_occ, int ncas, const int32_t* kap_row,
                                           const int32_t* kap_col, int n_kappa, int batch, double* work,
                                           double* H_matrix, unsigned eri_flags, oovqe_stream_t stream)
{
    const long na2 = (long)ncas * ncas;
    return orbital_hessian_batched(g_ao, h_ao, C, gamma, na2, Gamma, na2 * na2, fock, N, n_occ, ncas, kap_row,
                                   kap_col, n_kappa, batch, work, H_matrix, n_kappa,
                                   (long)n_kappa * n_kappa, nullptr, eri_flags, stream);
}

// ------------------------------------------------------------------------------------------------------
// configs[3]'s unit of work for a batch of geometries in ONE call: energy, full gradient and the full
// (n_theta + n_kappa)^2 Hessian of every geometry (OO_pqc.full_gradient + full_hessian,
// oo_pqc.py:132-148), the geometry index being a grid dimension of every launch.
// ------------------------------------------------------------------------------------------------------
int oovqe_oo_eval_batched_impl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                               int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                               const double* C, const double* nuc_arr, int N, int n_occ, int ncas,
                               const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int derivatives,
                               int batch, double* work, double* out, unsigned eri_flags,
                               oovqe_stream_t stream, const double* g_packed, double* fock,
                               const double* T2_ready = nullptr, hipEvent_t rdm_event = nullptr);
int oovqe_circuit_hessian_batched_impl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                       int n_gates, int n_qubits, int ncas, uint32_t init_index,
                                       const double* c1, const double* c2, long c1_bs, long c2_bs,
                                       const int32_t* pairs, int n_pairs, int batch, double* work, double* H,
                                       long ldh, long h_bs, oovqe_stream_t stream);

static bool hess_circuit_own_block(int n_qubits) { return n_qubits <= 10; }

extern "C" int64_t oovqe_oo_hessian_work_size(int n_theta, int n_gates, int n_qubits, int N, int n_occ,
                                              int ncas, int n_pairs)
{
    // per geometry: evaluation workspace | fock | max(circuit-Hessian, orbital-Hessian) workspace
    const int64_t ev = oovqe_oo_eval_work_size(n_theta, n_gates, n_qubits, N, n_occ, ncas, 1);
    const int64_t ch = oovqe_circuit_hessian_work_size(n_theta, n_qubits, ncas, n_pairs);
    const int64_t oh = oovqe_orbital_hessian_work_size(N, n_occ, ncas);
    // (small circuits: the circuit Hessian has a block of its own behind the shared one -- it runs BESIDE the
    // orbital Hessian's chains, oovqe_oo_hessian_batch)
    return ev + (int64_t)N * N + (ch > oh ? ch : oh) + (hess_circuit_own_block(n_qubits) ? ch : 0);
}

extern "C" int oovqe_oo_hessian_batch(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                      int n_gates, int n_qubits, uint32_t init_index, const double* g_ao,
                                      const double* h_ao, const double* C, const double* nuc, int N,
                                      int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                      int n_kappa, const int32_t* pairs, int n_pairs, int batch,
                                      double* work, double* out, double* hessian, unsigned eri_flags,
                                      const double* g_packed, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && g_ao && h_ao && C && nuc && work && out && hessian && pairs,
                  "oo_hessian_batch: null pointer");
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535 && n_theta >= 1 && n_kappa >= 1, "oo_hessian_batch: sizes");
    OOVQE_REQUIRE(n_pairs == n_theta * (n_theta + 1) / 2, "oo_hessian_batch: pairs must list every j <= k");
    const size_t nb = (size_t)batch;
    const long na2 = (long)ncas * ncas, na4 = na2 * na2;
    const int nvec = 1 + n_theta;
    const int64_t ev = oovqe_oo_eval_work_size(n_theta, n_gates, n_qubits, N, n_occ, ncas, 1);
    double* ev_work = work;                              // [G * ev]: starts with gamma [G][nvec][a^2], Gamma [G][nvec][a^4]
    double* fock = ev_work + nb * ev;                    // [G][N][N]
    double* hs_work = fock + nb * N * N;                 // circuit / orbital Hessian scratch, one after the other
    int rc;
    // 0. stage 1 of the integrals for the orbital Hessian first (T2[p,q,y,z] and the quarter transform Vk): the
    //    evaluation's packed-triangle path takes its J from that T2 instead of streaming the integrals again
    //    (p <-> q symmetric integrals, N <= 48, M <= 16: the shapes that path serves; option hess_own_stage1 = 1
    //    keeps the two passes apart)
    const int M = n_occ + ncas;
    const bool share = (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) != 0 && N <= 48 && M <= 16 &&
                       oovqe_opt(OOVQE_OPT_HESS_OWN_STAGE1) == 0;
    if (share && (rc = orbital_hessian_batched(g_ao, h_ao, C, nullptr, 0, nullptr, 0, nullptr, N, n_occ, ncas,
                                               kap_row, kap_col, n_kappa, batch, hs_work, nullptr, 0, 0, nullptr,
                                               eri_flags, stream, 1)))
        return rc;
    // The call is a graph, not a chain: behind stage 1 the evaluation (j_from_t2 -> q -> x / p -> n -> panels ->
    // assembly), the one-electron + J-type integrals, the circuit Hessian and the cross block (this stream) and the
    // K-type integrals + the assembly of the orbital block (the library's internal stream) do not depend on each
    // other.  Run one after the other they are 22 launches of 5-37 us at 8 geometries, 230 us.  Forked from and joined
    // to the caller's stream in here.
    hipStream_t st = (hipStream_t)stream;
    HessFork fork{nullptr, nullptr, nullptr, nullptr};
    bool forked = false;
    if (share && hess_circuit_own_block(n_qubits)) {
        fork.sK = oovqe_internal_stream(0);
        hipEvent_t ev_a = oovqe_internal_event();
        fork.rdm_ready = oovqe_internal_event();
        fork.rest_ready = oovqe_internal_event();
        fork.done = oovqe_internal_event();
        if (fork.sK && ev_a && fork.rdm_ready && fork.rest_ready && fork.done) {
            OOVQE_CHECK_HIP(hipEventRecord(ev_a, st), "oo_hessian_batch: hipEventRecord");
            OOVQE_CHECK_HIP(hipStreamWaitEvent(fork.sK, ev_a, 0), "oo_hessian_batch: hipStreamWaitEvent");
            forked = true;
            // the K-type chain needs stage 1 only: enqueued first
            if ((rc = orbital_hessian_batched(g_ao, h_ao, C, nullptr, 0, nullptr, 0, nullptr, N, n_occ, ncas, kap_row,
                                              kap_col, n_kappa, batch, hs_work, nullptr, 0, 0, nullptr, eri_flags,
                                              stream, 3, &fork)))
                return rc;
        }
    }
    // 1. circuit + tangents -> RDM sets -> CAS path: E, dE/dtheta, dE/dkappa, d^2E/dkappa dtheta, c1, c2, F
    if ((rc = oovqe_oo_eval_batched_impl(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, nuc,
                                         N, n_occ, ncas, kap_row, kap_col, n_kappa, 1, batch, ev_work, out,
                                         eri_flags, stream, g_packed, fock, share ? hs_work : nullptr,
                                         forked ? fork.rdm_ready : nullptr)))
        return rc;
    const long out_stride = (long)oovqe_oo_eval_out_size(n_theta, n_kappa, ncas, 1);
    const long n = (long)n_theta + n_kappa;
    const double* gvec = out + 2 + n_theta;
    const double* c1 = gvec + (size_t)nvec * n_kappa;
    const double* c2 = c1 + na2;
    // 2. kappa-kappa block (bottom right) from RDM set 0 of every geometry and its Fock matrix (before the
    //    circuit Hessian: that one reuses the scratch in which T2 / Vk of step 0 live)
    const double* gamma = ev_work;
    const double* Gamma = gamma + nb * nvec * na2;
    if ((rc = orbital_hessian_batched(g_ao, h_ao, C, gamma, (long)nvec * na2, Gamma, (long)nvec * na4, fock, N,
                                      n_occ, ncas, kap_row, kap_col, n_kappa, batch, hs_work,
                                      hessian + (size_t)n_theta * n + n_theta, n, n * n, nullptr, eri_flags, stream,
                                      share ? 2 : 0, forked ? &fork : nullptr)))
        return rc;
    // 3. theta-theta block (top left); beside the orbital block it works in a block of its own
    double* ch_work = hs_work;
    if (forked) {
        const int64_t ch = oovqe_circuit_hessian_work_size(n_theta, n_qubits, ncas, n_pairs);
        const int64_t oh = oovqe_orbital_hessian_work_size(N, n_occ, ncas);
        ch_work = hs_work + nb * (size_t)(ch > oh ? ch : oh);
    }
    if ((rc = oovqe_circuit_hessian_batched_impl(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index, c1,
                                                 c2, out_stride, out_stride, pairs, n_pairs, batch, ch_work,
                                                 hessian, n, n * n, stream)))
        return rc;
    // 4. kappa-theta blocks
    {
        const long total = (long)n_theta * n_kappa;
        const unsigned nbk = (unsigned)((total + 255) / 256);
        hess_cross_block_kernel<<<dim3(nbk, batch), 256, 0, (hipStream_t)stream>>>(gvec, out_stride, n_theta,
                                                                                  n_kappa, hessian, n, n * n);
        OOVQE_CHECK_LAUNCH("oo_hessian_batch/cross");
    }
    if (forked) OOVQE_CHECK_HIP(hipStreamWaitEvent(st, fork.done, 0), "oo_hessian_batch: hipStreamWaitEvent");
    return 0;
}
