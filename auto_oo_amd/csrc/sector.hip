// Particle-number-sector statevector engine with reverse-mode (adjoint) theta-gradients.
//
// UCCD / kUpCCD circuits conserve N_alpha and N_beta, so |psi(theta)> lives in the
// (N_alpha, N_beta) sector of the 2^n register: C(a,N_alpha) * C(a,N_beta) determinants
// (4 900 of 65 536 for CAS(8e,8o)).  The whole sector vector (39 KB) fits
// one workgroup's LDS, so a full kUpCCD circuit (56..112 Givens passes, reference
// src/auto_oo/ansatze/kUpCCD.py:118-130) runs in ONE launch without touching HBM between gates,
// one workgroup per batch element (theta set / geometry).
//
// Compressed index c = ia * nb + ib  <->  (alpha string, beta string)  <->  full basis index x
// (spin orbital 2p = alpha_p = qubit 2p = bit n-1-2p; strings keep orbital p at bit a-1-p):
//     x = spread(alpha) << 1 | spread(beta)
// Gates keep their full-register masks (include/oovqe.h, oovqe_gate_t): a thread owning
// determinant c applies the 2x2 rotation iff (x & (hi|lo)) == hi, partner = rank(x ^ (hi|lo)).
//
// Reverse mode (what torch autograd does for the reference, src/auto_oo/oo_pqc.py:86-95):
//   E(theta) = c0 + sum c1_pq gam_pq + sum c2_pqrs Gam_pqrs = c0 + psi^T Hop psi
//   lambda   = (Hop + Hop^T) psi                                   (sector_lambda_kernel)
//   walking the gates backwards:  dE/dtheta_k = (sign_k/2) lambda^T A_k psi,
//                                 psi <- U_k^T psi, lambda <- U_k^T lambda   (sector_adjoint_kernel)
// i.e. 2 vectors and 2*n_gates Givens passes instead of n_theta tangent states.
#include "common.h"

int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);

namespace {

constexpr int SEC_THREADS = 1024;

struct Sector {
    const uint32_t* unrank_a;   // [na]  alpha strings (orbital p at bit a-1-p)
    const uint32_t* unrank_b;   // [nb]
    const int32_t* rank_a;      // [2^a] string -> index or -1
    const int32_t* rank_b;      // [2^a]
    int na, nb, ncas;
};

__device__ __forceinline__ uint32_t spread16(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__device__ __forceinline__ uint32_t compact16(uint32_t v)
{
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}

__device__ __forceinline__ uint32_t sec_full(const Sector& s, int c)
{
    const int ia = c / s.nb, ib = c - ia * s.nb;
    return (spread16(s.unrank_a[ia]) << 1) | spread16(s.unrank_b[ib]);
}

__device__ __forceinline__ int sec_rank(const Sector& s, uint32_t x)
{
    return s.rank_a[compact16(x >> 1)] * s.nb + s.rank_b[compact16(x)];
}

// Stage the string tables and the full basis index of every determinant in LDS (the gate loops
// touch them 5 times per gate per thread; from global memory that is an L2 round trip each).
// `buf` must hold na + nb + 2*2^ncas + na*nb 32-bit words.  Returns the sector with LDS tables.
__device__ Sector sec_stage_lds(const Sector& g, uint32_t* buf, uint32_t** xfull, int nthreads)
{
    const int ns = 1 << g.ncas, Dc = g.na * g.nb;
    uint32_t* ua = buf;
    uint32_t* ub = ua + g.na;
    int32_t* ra = reinterpret_cast<int32_t*>(ub + g.nb);
    int32_t* rb = ra + ns;
    uint32_t* xf = reinterpret_cast<uint32_t*>(rb + ns);
    for (int i = threadIdx.x; i < g.na; i += nthreads) ua[i] = g.unrank_a[i];
    for (int i = threadIdx.x; i < g.nb; i += nthreads) ub[i] = g.unrank_b[i];
    for (int i = threadIdx.x; i < ns; i += nthreads) { ra[i] = g.rank_a[i]; rb[i] = g.rank_b[i]; }
    for (int d = threadIdx.x; d < Dc; d += nthreads) xf[d] = sec_full(g, d);
    Sector l = g;
    l.unrank_a = ua; l.unrank_b = ub; l.rank_a = ra; l.rank_b = rb;
    *xfull = xf;
    return l;
}

__host__ __device__ inline size_t sec_lds_words(int na, int nb, int ncas)
{
    return (size_t)na + nb + 2 * ((size_t)1 << ncas) + (size_t)na * nb;
}

// one Givens pass on a sector vector; transpose = apply U^T (adjoint sweep).  Up to SEC_MAXIT
// determinants per thread, processed as one unrolled batch so that the dependent LDS lookups
// (full index -> partner rank -> amplitudes) of different determinants overlap.
constexpr int SEC_MAXIT = 8;   // Dc <= SEC_MAXIT * SEC_THREADS (checked on the host)

template <int MAXIT>
__device__ __forceinline__ void sec_gate(double* st, const Sector& s, const uint32_t* xfull, int Dc,
                                         const oovqe_gate_t& g, double c, double sn, bool transpose)
{
    const uint32_t fm = g.mask_hi | g.mask_lo;
    const double sg = transpose ? -sn : sn;
    int e[MAXIT];
    double pi[MAXIT];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        const int d = threadIdx.x + i * SEC_THREADS;
        e[i] = -1;
        if (d < Dc) {
            const uint32_t x = xfull[d];
            if ((x & fm) == g.mask_hi) {
                e[i] = sec_rank(s, x ^ fm);
                pi[i] = (__popc(x & g.mask_par) & 1) ? -sg : sg;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        if (e[i] >= 0) {
            const int d = threadIdx.x + i * SEC_THREADS;
            const double ax = st[d], ay = st[e[i]];
            st[d] = c * ax + pi[i] * ay;
            st[e[i]] = c * ay - pi[i] * ax;
        }
    }
}

// ---- forward circuit: grid = batch -------------------------------------------------------------
template <int MAXIT>
__global__ __launch_bounds__(SEC_THREADS)
void sector_circuit_kernel(const double* __restrict__ theta, int n_theta,
                           const oovqe_gate_t* __restrict__ gates, int n_gates, Sector s,
                           uint32_t init_index, double* __restrict__ psi_c)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb;
    double* st = lds;                                   // [Dc]
    double* cs = st + Dc;                               // [n_gates][2]
    oovqe_gate_t* gl = reinterpret_cast<oovqe_gate_t*>(cs + 2 * n_gates);
    uint32_t* xfull;
    const Sector sg = s;
    s = sec_stage_lds(sg, reinterpret_cast<uint32_t*>(gl + n_gates), &xfull, SEC_THREADS);
    const int tid = threadIdx.x, b = blockIdx.x;
    const double* th = theta + (size_t)b * n_theta;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gates);
        uint32_t* dst = reinterpret_cast<uint32_t*>(gl);
        for (int i = tid; i < n_gates * (int)(sizeof(oovqe_gate_t) / 4); i += SEC_THREADS) dst[i] = src[i];
        for (int g = tid; g < n_gates; g += SEC_THREADS) {
            const int ti = gates[g].theta_idx;
            double sn = 0.0, c = 1.0;
            if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
            cs[2 * g] = c;
            cs[2 * g + 1] = sn;
        }
        const int c0 = sec_rank(sg, init_index);
        for (int d = tid; d < Dc; d += SEC_THREADS) st[d] = (d == c0) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int g = 0; g < n_gates; ++g) {
        if (gl[g].theta_idx < 0) continue;
        sec_gate<MAXIT>(st, s, xfull, Dc, gl[g], cs[2 * g], cs[2 * g + 1], false);
        __syncthreads();
    }
    for (int d = tid; d < Dc; d += SEC_THREADS) psi_c[(size_t)b * Dc + d] = st[d];
}

// d^order/dtheta^order of one Givens pass applied to a sector vector (order = 1, 2): on the pairs the gate
// rotates the 2 x 2 block [c, s; -s, c] (c, s = cos, sin of sign theta / 2) becomes its derivative --
// (sign / 2) [-s, c; -c, -s], then (1 / 4) [-c, -s; s, -c] -- and every amplitude the gate leaves alone is
// annihilated (the derivative of an identity block).  `c`, `sn` arrive already differentiated.
template <int MAXIT>
__device__ __forceinline__ void sec_gate_deriv(double* st, const Sector& s, const uint32_t* xfull, int Dc,
                                               const oovqe_gate_t& g, double c, double sn)
{
    const uint32_t fm = g.mask_hi | g.mask_lo;
    int e[MAXIT];
    double pi[MAXIT];
    bool idle[MAXIT];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        const int d = threadIdx.x + i * SEC_THREADS;
        e[i] = -1;
        idle[i] = false;
        if (d < Dc) {
            const uint32_t x = xfull[d];
            if ((x & fm) == g.mask_hi) {
                e[i] = sec_rank(s, x ^ fm);
                pi[i] = (__popc(x & g.mask_par) & 1) ? -sn : sn;
            } else if ((x & fm) != g.mask_lo) {
                idle[i] = true;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
        const int d = threadIdx.x + i * SEC_THREADS;
        if (e[i] >= 0) {
            const double ax = st[d], ay = st[e[i]];
            st[d] = c * ax + pi[i] * ay;
            st[e[i]] = c * ay - pi[i] * ax;
        } else if (idle[i]) {
            st[d] = 0.0;
        }
    }
}

// The circuit with up to two of its gates differentiated: output o of geometry b is
//   prod_g G_g^(m_g)(theta) |init>,  m_g = how often gate g appears in deriv[o] = (g_a, g_b)  (-1: none),
// i.e. a first tangent d psi / d theta_j (g_a = the gate of theta_j, g_b = -1), a second tangent
// d^2 psi / d theta_j d theta_k (both set; g_a == g_b: the second derivative of that gate) or psi itself,
// for circuits in which every parameter drives ONE gate (UCCD / UCCSD / kUpCCD: one Givens pass per
// excitation).  grid = (batch, n_out); one sector vector in LDS per workgroup, as in the plain kernel.
template <int MAXIT>
__global__ __launch_bounds__(SEC_THREADS)
void sector_circuit_deriv_kernel(const double* __restrict__ theta, int n_theta,
                                 const oovqe_gate_t* __restrict__ gates, int n_gates, Sector s,
                                 uint32_t init_index, const int32_t* __restrict__ deriv, int n_out,
                                 double* __restrict__ psi_c)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb;
    double* st = lds;                                   // [Dc]
    double* cs = st + Dc;                               // [n_gates][2]
    oovqe_gate_t* gl = reinterpret_cast<oovqe_gate_t*>(cs + 2 * n_gates);
    uint32_t* xfull;
    const Sector sg = s;
    s = sec_stage_lds(sg, reinterpret_cast<uint32_t*>(gl + n_gates), &xfull, SEC_THREADS);
    const int tid = threadIdx.x, b = blockIdx.x, o = blockIdx.y;
    const double* th = theta + (size_t)b * n_theta;
    const int ga = deriv[2 * o], gb = deriv[2 * o + 1];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gates);
        uint32_t* dst = reinterpret_cast<uint32_t*>(gl);
        for (int i = tid; i < n_gates * (int)(sizeof(oovqe_gate_t) / 4); i += SEC_THREADS) dst[i] = src[i];
        for (int g = tid; g < n_gates; g += SEC_THREADS) {
            const int ti = gates[g].theta_idx;
            double sn = 0.0, c = 1.0;
            if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
            const int order = (g == ga ? 1 : 0) + (g == gb ? 1 : 0);
            const double h = 0.5 * (double)gates[g].sign;
            if (order == 1) { const double c1 = -h * sn, s1 = h * c; c = c1; sn = s1; }
            if (order == 2) { c *= -0.25; sn *= -0.25; }
            cs[2 * g] = c;
            cs[2 * g + 1] = sn;
        }
        const int c0 = sec_rank(sg, init_index);
        for (int d = tid; d < Dc; d += SEC_THREADS) st[d] = (d == c0) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int g = 0; g < n_gates; ++g) {
        if (gl[g].theta_idx < 0) continue;
        if (g == ga || g == gb) sec_gate_deriv<MAXIT>(st, s, xfull, Dc, gl[g], cs[2 * g], cs[2 * g + 1]);
        else sec_gate<MAXIT>(st, s, xfull, Dc, gl[g], cs[2 * g], cs[2 * g + 1], false);
        __syncthreads();
    }
    for (int d = tid; d < Dc; d += SEC_THREADS) psi_c[((size_t)b * n_out + o) * Dc + d] = st[d];
}

// sector vector -> dense 2^n vector (zeros outside the sector)
__global__ void sector_to_dense_kernel(const double* __restrict__ psi_c, Sector s, uint32_t D,
                                       double* __restrict__ psi)
{
    const int Dc = s.na * s.nb;
    const size_t b = blockIdx.y;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < D; x += gridDim.x * blockDim.x) {
        const int ia = s.rank_a[compact16(x >> 1)], ib = s.rank_b[compact16(x)];
        psi[b * D + x] = (ia >= 0 && ib >= 0) ? psi_c[b * Dc + ia * s.nb + ib] : 0.0;
    }
}

// ---- E_pq applications in the sector: V[vec][pq][c] ---------------------------------------------
__device__ __forceinline__ double sec_epq(const double* __restrict__ src, const Sector& s, int n,
                                          int p, int q, uint32_t x, int c)
{
    double acc = 0.0;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int P = 2 * p + sp, Q = 2 * q + sp;
        const uint32_t bP = 1u << (n - 1 - P), bQ = 1u << (n - 1 - Q);
        if (p == q) {
            if (x & bP) acc += src[c];
        } else if ((x & bP) && !(x & bQ)) {
            const uint32_t hi = bP > bQ ? bP : bQ, lo = bP > bQ ? bQ : bP;
            const uint32_t between = (hi - 1u) & ~((lo << 1) - 1u);
            const double sgn = (__popc(x & between) & 1) ? -1.0 : 1.0;
            acc += sgn * src[sec_rank(s, x ^ (bP | bQ))];
        }
    }
    return acc;
}

// grid: (ceil(Dc/256), a^2, nvec_total); vec [nvec_total][Dc] -> V [nvec_total][a^2][Dc]
__global__ __launch_bounds__(256)
void sector_epq_kernel(const double* __restrict__ vec, Sector s, double* __restrict__ V)
{
    const int Dc = s.na * s.nb, n = 2 * s.ncas, na2 = s.ncas * s.ncas;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Dc) return;
    const int pq = blockIdx.y, p = pq / s.ncas, q = pq - p * s.ncas;
    const size_t v = blockIdx.z;
    const uint32_t x = sec_full(s, c);
    V[(v * na2 + pq) * Dc + c] = sec_epq(vec + v * Dc, s, n, p, q, x, c);
}

// The same with ONE workgroup per (256 determinants, state) forming all a^2 vectors: the source
// vector and the rank tables are staged in LDS once and each thread walks the (p, q) pairs of its
// determinant -- 20 workgroups per state instead of 1 280 one-element-per-thread workgroups, and
// the gathers E_pq needs come from LDS instead of L2 (batch 256: 327 -> ~150 us, the rate at which
// the 642 MB of V can be written).  grid: (ceil(Dc/256), nvec_total)
__global__ __launch_bounds__(256)
void sector_epq_rows_kernel(const double* __restrict__ vec, Sector s, double* __restrict__ V)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb, n = 2 * s.ncas, a = s.ncas, na2 = a * a;
    const int ns = 1 << a;
    double* src = lds;                                           // [Dc]
    int32_t* ra = reinterpret_cast<int32_t*>(src + Dc + (Dc & 1));
    int32_t* rb = ra + ns;
    const size_t v = blockIdx.y;
    for (int i = threadIdx.x; i < Dc; i += 256) src[i] = vec[v * Dc + i];
    for (int i = threadIdx.x; i < ns; i += 256) { ra[i] = s.rank_a[i]; rb[i] = s.rank_b[i]; }
    const Sector sg = s;
    s.rank_a = ra;
    s.rank_b = rb;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Dc) return;
    const uint32_t x = sec_full(sg, c);
    double* out = V + v * (size_t)na2 * Dc + c;
    for (int p = 0; p < a; ++p)
#pragma unroll 4
        for (int q = 0; q < a; ++q) out[(size_t)(p * a + q) * Dc] = sec_epq(src, s, n, p, q, x, c);
}

// ---- RDM Gram on the f64 MFMA -----------------------------------------------------------------
// G[m][n] = sum_c A[m][c] B[n][c];  A rows m < a^2: V[qp] (m = pq), row a^2: psi;  B rows: V[rs].
// grid: (MT*NT tiles, batch, ksplit); the 4 waves of a block split their c range again; operands
// are fetched 8 k-steps at a time so the L2 latency is paid once per 8 MFMAs; partial results of
// the ksplit slices land in R[split] and are summed in fixed order by the finish kernel.
__global__ __launch_bounds__(256)
void sector_gram_kernel(const double* __restrict__ psi_c, const double* __restrict__ V, int ncas,
                        int Dc, int batch, double* __restrict__ R)
{
    __shared__ double red[4][256];
    const int na2 = ncas * ncas, nrow = na2 + 1;
    const int MT = (nrow + 15) / 16, NT = (na2 + 15) / 16;
    const int tile = blockIdx.x, mt = tile / NT, nt = tile - mt * NT;
    const size_t b = blockIdx.y;
    const int split = blockIdx.z, nsplit = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const double* Vb = V + b * (size_t)na2 * Dc;
    const int m = mt * 16 + lr, nn = nt * 16 + lr;
    const double* arow = psi_c + b * (size_t)Dc;    // dummy valid pointer for padded rows
    double amask = 0.0, bmask = 0.0;
    if (m < na2) {
        const int p = m / ncas, q = m - p * ncas;
        arow = Vb + (size_t)(q * ncas + p) * Dc;
        amask = 1.0;
    } else if (m == na2) {
        amask = 1.0;
    }
    const double* brow = Vb;
    if (nn < na2) { brow = Vb + (size_t)nn * Dc; bmask = 1.0; }
    const int ksteps = (Dc + 3) / 4;
    const int nslice = 4 * nsplit;
    const int per = (ksteps + nslice - 1) / nslice;
    const int sl = split * 4 + wave;
    const int k0 = sl * per, k1 = (k0 + per < ksteps) ? k0 + per : ksteps;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int ks = k0; ks < k1; ks += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (ks + u) + lq;
            const int cc = c < Dc ? c : Dc - 1;
            const double ok = (ks + u < k1 && c < Dc) ? 1.0 : 0.0;
            av[u] = arow[cc] * (ok * amask);
            bv[u] = brow[cc] * (ok * bmask);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma_f64(av[u], bv[u], acc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(lq + 4 * i) * 16 + lr] = acc[i];
    __syncthreads();
    const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const int row = mt * 16 + tid / 16, col = nt * 16 + (tid & 15);
    R[(((size_t)split * batch + b) * (MT * 16) + row) * (NT * 16) + col] = v;
}

// The same Gram with ONE workgroup per (state, c-slice) forming all MT x NT tiles at once: every row
// of V is read once per workgroup (the tile-per-workgroup kernel above re-reads each row MT or NT
// times, and its tiles of one state land on different XCDs: at batch 256 the 642 MB of V came from
// HBM / Infinity Cache ~8 times, 965 us where the bytes take 110).  A lane fetches 32 contiguous
// bytes (4 determinants) of its row per 16-determinant chunk -- four lanes cover a 128-byte line --
// and k-step j of the chunk takes determinant 16 ch + 4 lq + j from lane group lq: any assignment
// of determinants to k-steps gives the same sum as long as A and B use the same one.
// PSI_VALU (a^2 a multiple of 16, e.g. 8 active orbitals): the extra row <psi| V_rs> = gamma_rs would
// cost a whole row of tiles with one useful row in sixteen (MT = 5 instead of 4: a fifth of the
// MFMAs); it is formed on the vector ALUs instead, from the B fragments the lanes hold anyway.
template <int MT, int NT, bool PSI_VALU>
__global__ __launch_bounds__(512)
void sector_gram_rows_kernel(const double* __restrict__ psi_c, const double* __restrict__ V, int ncas,
                             int Dc, int batch, int MTR, double* __restrict__ R)
{
    // MTR: row tiles of the result buffer R; blockIdx.z: which MT row tiles this workgroup forms
    // (two half-height workgroups per state keep 4 waves per SIMD resident at 128 VGPRs each)
    __shared__ double red[8][256];
    const int mt0 = blockIdx.z * MT;
    const int na2 = ncas * ncas;
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const double* Vb = V + b * (size_t)na2 * Dc;
    const double* arow[MT];
    double amask[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = (mt0 + mt) * 16 + lr;
        arow[mt] = psi_c + b * (size_t)Dc;          // row a^2 = psi; also the dummy of padded rows
        amask[mt] = m <= na2 ? 1.0 : 0.0;
        if (m < na2) {
            const int p = m / ncas, q = m - p * ncas;
            arow[mt] = Vb + (size_t)(q * ncas + p) * Dc;
        }
    }
    const double* brow[NT];
    double bmask[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nn = nt * 16 + lr;
        brow[nt] = Vb + (size_t)(nn < na2 ? nn : 0) * Dc;
        bmask[nt] = nn < na2 ? 1.0 : 0.0;
    }
    const int nchunk = (Dc + 15) / 16;
    const int nslice = 8 * nsplit;
    const int per = (nchunk + nslice - 1) / nslice;
    const int sl = split * 8 + wave;
    const int ch0 = sl * per, ch1 = (ch0 + per < nchunk) ? ch0 + per : nchunk;
    d4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
    double gpart[NT];                                 // PSI_VALU: this lane's share of <psi| V_row>
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) gpart[nt] = 0.0;
    const double* prow = psi_c + b * (size_t)Dc;
    for (int ch = ch0; ch < ch1; ++ch) {
        const int c0 = 16 * ch + 4 * lq;
        // Dc is a multiple of 4 for every sector with an even number of strings per spin, but not in
        // general: positions past the end are clamped and masked one by one
        double km[4];
        int cc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { cc[j] = c0 + j < Dc ? c0 + j : Dc - 1; km[j] = c0 + j < Dc ? 1.0 : 0.0; }
        const bool whole = c0 + 3 < Dc && (Dc & 1) == 0;      // 16-byte loads need an even row pitch
        double bv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (whole) {
                const d2 lo = *reinterpret_cast<const d2*>(brow[nt] + c0), hi = *reinterpret_cast<const d2*>(brow[nt] + c0 + 2);
                bv[nt][0] = lo.x * bmask[nt]; bv[nt][1] = lo.y * bmask[nt];
                bv[nt][2] = hi.x * bmask[nt]; bv[nt][3] = hi.y * bmask[nt];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[nt][j] = brow[nt][cc[j]] * (km[j] * bmask[nt]);
            }
        }
        if (PSI_VALU && blockIdx.z == 0) {
            double pv[4];
            if (whole) {
                const d2 lo = *reinterpret_cast<const d2*>(prow + c0), hi = *reinterpret_cast<const d2*>(prow + c0 + 2);
                pv[0] = lo.x; pv[1] = lo.y; pv[2] = hi.x; pv[3] = hi.y;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[j] = prow[cc[j]] * km[j];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) gpart[nt] += pv[j] * bv[nt][j];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            double av[4];
            if (whole) {
                const d2 lo = *reinterpret_cast<const d2*>(arow[mt] + c0), hi = *reinterpret_cast<const d2*>(arow[mt] + c0 + 2);
                av[0] = lo.x * amask[mt]; av[1] = lo.y * amask[mt];
                av[2] = hi.x * amask[mt]; av[3] = hi.y * amask[mt];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = arow[mt][cc[j]] * (km[j] * amask[mt]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_f64(av[j], bv[nt][j], acc[mt][nt]);
        }
    }
    // the 8 waves' partial tiles are summed through LDS in fixed order, tile by tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][(lq + 4 * i) * 16 + lr] = acc[mt][nt][i];
            __syncthreads();
            if (tid < 256) {
                double v = red[0][tid];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += red[w][tid];
                const int row = (mt0 + mt) * 16 + tid / 16, col = nt * 16 + (tid & 15);
                R[(((size_t)split * batch + b) * (MTR * 16) + row) * (NT * 16) + col] = v;
            }
            __syncthreads();
        }
    if (PSI_VALU && blockIdx.z == 0) {
        // row a^2 of R: lane (lq, lr) holds the share of row nt*16 + lr over its determinants; the
        // four lane groups and the eight waves meet in LDS, summed in fixed order
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            red[wave][lq * 16 + lr] = gpart[nt];
            __syncthreads();
            if (tid < 16) {
                double v = 0.0;
                for (int w = 0; w < 8; ++w)
                    for (int g = 0; g < 4; ++g) v += red[w][g * 16 + tid];
                R[(((size_t)split * batch + b) * (MTR * 16) + na2) * (NT * 16) + nt * 16 + tid] = v;
            }
            __syncthreads();
        }
    }
}

// ---- round 3: the E_pq vectors never leave the chip --------------------------------------------------
// V = (E_pq psi) is 64 x 4 900 doubles per CAS(8e,8o) state: written once and read once it was 1.28 GB of the
// RDM stage's traffic at batch 256 (and as much again for the adjoint).  But both consumers are LOCAL in the
// determinant index -- the Gram sums over it, W = Ms^T V acts on the (p,q) index -- so a workgroup that holds
// psi (39 KB) in LDS can form V for a chunk of SEC_CH determinants in LDS (a^2 x SEC_CH doubles), use it and
// drop it.  Both kernels below: grid (batch, nsplit), 512 threads, the chunks of a state dealt to the
// nsplit workgroups; four threads per determinant build the chunk (a^2 / 4 operators each).
constexpr int SEC_CH = 128;                  // determinants per chunk
// LDS pitch of a V row, chosen per kernel for its MFMA operand reads (64 banks of 4 bytes, 32 lanes per pass):
//   Gram: a lane reads V[row = tile row lr][column cc + lq]: rows 4 banks apart  -> 128 + 2 doubles
//   W:    a lane reads V[row = k-step row lq][column 16 ct + lr]: rows 32 banks apart -> 128 + 16 doubles
constexpr int SEC_CHP_GRAM = SEC_CH + 2;
constexpr int SEC_CHP_W = SEC_CH + 16;
constexpr int SEC_CHP_MAX = SEC_CHP_W;

// Excitation tables of the strings (built by every workgroup at its start, a few hundred entries per thread
// once): E_pq = E^alpha_pq + E^beta_pq, and on the determinant (ia, ib)
//   (E^alpha_pq v)[ia, ib] = (-1)^(own_a(ia) + cross_b(ib)) v[src_a(ia), ib]     if valid_a(ia)
// where src_a = the alpha string with the electron moved back from p to q, own_a the parity of the alpha
// electrons strictly between p and q, cross_b the parity of the beta electrons the Jordan-Wigner string of an
// alpha excitation crosses (orbitals min(p,q) .. max(p,q) - 1; for a beta excitation the alpha electrons in
// min + 1 .. max) -- spin orbital 2p = alpha_p, 2p + 1 = beta_p, utils/active_space.py:29-83.  One 16-bit word
// per (string, pq): source index (11 bits) | valid | own parity | cross parity.  The per-determinant bit
// arithmetic of sec_epq (two rank look-ups, two popcounts, four branches per operator) becomes two table
// reads: the chunk build went from 60 % of the fused kernels' time to a tenth.
constexpr int SEC_TAB_MAXSTR = 2048;

__host__ __device__ inline size_t sec_fused_lds_bytes(int na, int nb, int ncas)
{
    const size_t Dc = (size_t)na * nb, na2 = (size_t)ncas * ncas;
    size_t bytes = (Dc + (Dc & 1) + na2 * SEC_CHP_MAX) * sizeof(double);
    bytes += (((size_t)na + nb) * na2 * sizeof(uint16_t) + 7) & ~(size_t)7;
    return bytes;
}

__device__ __forceinline__ uint32_t sec_orb_mask(int a, int r1, int r2)     // orbitals r1 .. r2 of a string
{
    return r2 < r1 ? 0u : (((1u << (r2 - r1 + 1)) - 1u) << (a - 1 - r2));
}

// tab[pq * nstr + (string index)]; is_alpha selects which cross range the string serves
__device__ void sec_build_table(const uint32_t* __restrict__ unrank, const int32_t* __restrict__ rank, int nstr,
                                int a, bool is_alpha, uint16_t* __restrict__ tab, int nthreads)
{
    const int na2 = a * a;
    for (int idx = threadIdx.x; idx < nstr * na2; idx += nthreads) {
        const int is = idx / na2, pq = idx - is * na2, p = pq / a, q = pq - p * a;
        const uint32_t st = unrank[is];
        const uint32_t bp = 1u << (a - 1 - p), bq = 1u << (a - 1 - q);
        const int lo = p < q ? p : q, hi = p < q ? q : p;
        uint32_t valid, src;
        if (p == q) { valid = (st & bp) ? 1u : 0u; src = (uint32_t)is; }
        else {
            valid = ((st & bp) && !(st & bq)) ? 1u : 0u;
            src = valid ? (uint32_t)rank[(st & ~bp) | bq] : 0u;
        }
        const uint32_t own = __popc(st & sec_orb_mask(a, lo + 1, hi - 1)) & 1u;
        // the range of THIS spin's electrons that an excitation of the OTHER spin crosses
        const uint32_t cross = is_alpha ? (__popc(st & sec_orb_mask(a, lo + 1, hi)) & 1u)
                                        : (__popc(st & sec_orb_mask(a, lo, hi - 1)) & 1u);
        // operator-major: the lanes of a wave hold consecutive beta strings and read one operator's word each
        // (string-major, 128 bytes apart, all 64 reads fell on two LDS banks)
        tab[pq * nstr + is] = (uint16_t)(src | (valid << 11) | (own << 12) | (cross << 13));
    }
}

// the two tables of a workgroup: copied from the per-circuit copy in memory (oovqe_sector_pairs leaves one behind
// its pair lists: [a^2][na] | [a^2][nb]) when the caller holds it, built from the strings otherwise
__device__ __forceinline__ void sec_tables(const Sector& s, const uint16_t* __restrict__ tabs_g, uint16_t* tabA,
                                           uint16_t* tabB, int nthreads)
{
    const int na2 = s.ncas * s.ncas;
    if (tabs_g) {
        for (int i = threadIdx.x; i < s.na * na2; i += nthreads) tabA[i] = tabs_g[i];
        for (int i = threadIdx.x; i < s.nb * na2; i += nthreads) tabB[i] = tabs_g[(size_t)s.na * na2 + i];
    } else {
        sec_build_table(s.unrank_a, s.rank_a, s.na, s.ncas, true, tabA, nthreads);
        sec_build_table(s.unrank_b, s.rank_b, s.nb, s.ncas, false, tabB, nthreads);
    }
}

// fills Vc[pq][0 .. SEC_CH) for the determinants c0 .. c0 + SEC_CH - 1 (zeros behind the sector)
template <int SEC_CHP>
__device__ __forceinline__ void sec_build_chunk(const double* __restrict__ src, int na, int nb, int a,
                                                const uint16_t* __restrict__ tabA,
                                                const uint16_t* __restrict__ tabB, int c0,
                                                double* __restrict__ Vc)
{
    const int Dc = na * nb, na2 = a * a;
    const int cl = threadIdx.x & (SEC_CH - 1), part = threadIdx.x / SEC_CH;       // 512 / 128 = 4 parts
    const int c = c0 + cl;
    const bool in = c < Dc;
    const int ia = in ? c / nb : 0, ib = in ? c - ia * nb : 0;
    const uint16_t* ta = tabA + ia;
    const uint16_t* tb = tabB + ib;
    const double* rowa = src + ia * nb;
    // eight operators at a time: the table reads, then the amplitude reads, then the stores (one operator
    // after the other is a chain of four dependent LDS round trips each)
    for (int pq0 = part; pq0 < na2; pq0 += 8 * (512 / SEC_CH)) {
        uint32_t ea[8], eb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pq = pq0 + u * (512 / SEC_CH);
            ea[u] = pq < na2 ? ta[pq * na] : 0u;
            eb[u] = pq < na2 ? tb[pq * nb] : 0u;
        }
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            va[u] = src[__umul24(ea[u] & 2047u, (unsigned)nb) + ib];      // (24-bit multiply: full rate)
            vb[u] = rowa[eb[u] & 2047u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pq = pq0 + u * (512 / SEC_CH);
            const bool sa = ((ea[u] >> 12) ^ (eb[u] >> 13)) & 1u, sb = ((eb[u] >> 12) ^ (ea[u] >> 13)) & 1u;
            const double xa = (ea[u] & 2048u) ? (sa ? -va[u] : va[u]) : 0.0;
            const double xb = (eb[u] & 2048u) ? (sb ? -vb[u] : vb[u]) : 0.0;
            if (pq < na2) Vc[pq * SEC_CHP + cl] = in ? xa + xb : 0.0;
        }
    }
}

// RDM Gram without V in memory: R as sector_gram_rows_kernel writes it (same finish kernel).
// a^2 a multiple of 16 (a = 4, 8): MT = NT = a^2 / 16, the <psi| V_rs> row on the vector ALUs.
template <int NT>
__global__ __launch_bounds__(512)
void sector_rdm_fused_kernel(const double* __restrict__ psi_c, Sector s, int batch, int MTR, double* __restrict__ R,
                             int probe, const uint16_t* __restrict__ tabs_g)
{
    extern __shared__ double lds[];
    __shared__ double red[8][256];
    const int Dc = s.na * s.nb, a = s.ncas, na2 = a * a;
    double* src = lds;                                            // [Dc]
    double* Vc = src + Dc + (Dc & 1);                             // [a^2][SEC_CHP]
    constexpr int SEC_CHP = SEC_CHP_GRAM;
    uint16_t* tabA = reinterpret_cast<uint16_t*>(Vc + (size_t)na2 * SEC_CHP_MAX);
    uint16_t* tabB = tabA + (size_t)s.na * na2;
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    for (int i = tid; i < Dc; i += 512) src[i] = psi_c[b * Dc + i];
    sec_tables(s, tabs_g, tabA, tabB, 512);
    // A row m' of the product is V[m'] as well (the same conflict-free operand reads as B); it stands for
    // the pair (p,q) = (m' % a, m' / a), i.e. V[(q,p)] is row pq of the Gram: the rows are permuted when R
    // is written

    d4 acc[NT][NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
    double gpart[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) gpart[nt] = 0.0;
    const int nchunk = (Dc + SEC_CH - 1) / SEC_CH;
    __syncthreads();
    for (int ch = split; ch < nchunk; ch += nsplit) {
        const int c0 = ch * SEC_CH;
        if (!(probe == 1 && ch != split)) sec_build_chunk<SEC_CHP>(src, s.na, s.nb, a, tabA, tabB, c0, Vc);
        __syncthreads();
        // the 32 k-steps of the chunk dealt to the 8 waves; k-step ks covers determinants 4 ks .. 4 ks + 3
        if (probe != 2)
#pragma unroll
        for (int u = 0; u < SEC_CH / 4 / 8; ++u) {
            const int cc = 4 * (wave + 8 * u) + lq;
            double bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = Vc[(nt * 16 + lr) * SEC_CHP + cc];
            const double pv = c0 + cc < Dc ? src[c0 + cc] : 0.0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gpart[nt] += pv * bv[nt];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)              // (A row tile mt holds the same V rows as B tile mt)
#pragma unroll
                for (int nt = mt; nt < NT; ++nt)         // V^T V is symmetric: the tiles mt <= nt only
                    acc[mt][nt] = mfma_f64(bv[mt], bv[nt], acc[mt][nt]);
        }
        __syncthreads();
    }
    // the 8 waves' partial tiles are summed through LDS in fixed order, tile by tile
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][(lq + 4 * i) * 16 + lr] = acc[mt][nt][i];
            __syncthreads();
            if (tid < 256) {
                double v = red[0][tid];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += red[w][tid];
                const int mrow = mt * 16 + tid / 16, col = nt * 16 + (tid & 15);
                const int row = (mrow % a) * a + mrow / a;            // product row (q,p) -> Gram row (p,q)
                double* Rb = R + ((size_t)split * batch + b) * (MTR * 16) * (NT * 16);
                Rb[(size_t)row * (NT * 16) + col] = v;
                // the mirror tile (nt, mt): product element (col, mrow), the same sum in the same order
                if (mt != nt) Rb[(size_t)((col % a) * a + col / a) * (NT * 16) + mrow] = v;
            }
            __syncthreads();
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        red[wave][lq * 16 + lr] = gpart[nt];
        __syncthreads();
        if (tid < 16) {
            double v = 0.0;
            for (int w = 0; w < 8; ++w)
                for (int gq = 0; gq < 4; ++gq) v += red[w][gq * 16 + tid];
            R[(((size_t)split * batch + b) * (MTR * 16) + na2) * (NT * 16) + nt * 16 + tid] = v;
        }
        __syncthreads();
    }
}

// ---- round 4: the Gram from ROW chunks in the sigma basis ----------------------------------------------------------
// sector_rdm_fused_kernel spends 103 of its 226 us (256 CAS(8e,8o) states) building V: ~20 vector instructions per
// element (two table words, two amplitudes, own and cross parities, two selects).  In the basis psi' = sigma psi
// (see the string-driven lambda below: sigma^2 = 1, so the Gram and <psi| V> do not change) the cross parities
// are gone, and with chunks of WHOLE alpha rows (all beta strings of 144 / NBp rows) a thread keeps its beta
// string: its beta-excitation words -- source column (or a zero column), sign -- are read ONCE into registers; the
// alpha words come from a 32-bit table that holds the byte offset of the source ROW (or of a zero row) and the
// sign bit, so that an element of V costs two LDS reads, two XORs, four integer operations and one add.
constexpr int SEC_RCHP = SEC_CH + 18;          // 146: V rows 36 banks apart, conflict-free Gram operand reads
static_assert((SEC_RCHP / 2) % 2 == 1 && SEC_RCHP % 2 == 0, "pitch / 2 must be odd");
constexpr int SEC_RCH = 144;                   // columns of a row chunk (whole alpha rows x padded nb)

__device__ __forceinline__ double sec_sigma(const Sector& s, int ia, int ib)
{
    const uint32_t sa = s.unrank_a[ia], sb = s.unrank_b[ib];
    uint32_t par = 0;
    for (int i = 1; i < s.ncas; ++i)
        if (sa & (1u << (s.ncas - 1 - i))) par ^= __popc(sb & sec_orb_mask(s.ncas, 0, i - 1)) & 1u;
    return par ? -1.0 : 1.0;
}

__host__ __device__ inline size_t sec_rdm_rows_lds_bytes(int na, int nb, int ncas)
{
    const size_t na2 = (size_t)ncas * ncas, NBp = ((size_t)nb + 8) & ~(size_t)7, NRC = SEC_RCH / NBp;
    if (NRC < 1) return (size_t)1 << 30;
    const size_t nchunk = (na + NRC - 1) / NRC;
    const size_t NR = nchunk * NRC > (size_t)na + 1 ? nchunk * NRC : (size_t)na + 1;
    return (NR * NBp + na2 * SEC_RCHP) * sizeof(double) + (na2 * (size_t)na * sizeof(uint32_t) + 7) / 8 * 8;
}

template <int NT>
__global__ __launch_bounds__(512)
void sector_rdm_rows_kernel(const double* __restrict__ psi_c, Sector s, int batch, int MTR, double* __restrict__ R,
                            int probe, const uint16_t* __restrict__ tabs_g)
{
    extern __shared__ double lds[];
    __shared__ double red[8][256];
    const int na = s.na, nb = s.nb, Dc = na * nb, a = s.ncas, na2 = a * a;
    const int NBp = (nb + 8) & ~7, NRC = SEC_RCH / NBp, CH = NRC * NBp, KSC = CH / 4;   // (>= 1 zero column behind nb)
    const int nchunk = (na + NRC - 1) / NRC;
    const int NR = nchunk * NRC > na + 1 ? nchunk * NRC : na + 1;     // rows >= na are zero
    double* P = lds;                                   // [NR][NBp]  sigma psi, zero padded
    double* Vc = P + (size_t)NR * NBp;                 // [a^2][SEC_RCHP]
    uint32_t* tA = reinterpret_cast<uint32_t*>(Vc + (size_t)na2 * SEC_RCHP);   // [a^2][na]
    uint16_t* tmp = reinterpret_cast<uint16_t*>(Vc);   // the 16-bit tables, before the first chunk is built
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    for (int i = tid; i < NR * NBp; i += 512) {
        const int r = i / NBp, c = i - r * NBp;
        P[i] = (r < na && c < nb) ? sec_sigma(s, r, c) * psi_c[b * Dc + r * nb + c] : 0.0;
    }
    // this thread's column of a chunk and its share of the operators
    constexpr int MAXOP = NT == 4 ? 24 : 8;
    const int NP = 512 / CH < 4 ? 512 / CH : 4;
    const int part = tid / CH, l = tid - part * CH;
    const int ial = l / NBp, ib = l - ial * NBp;
    const int opp = (na2 + NP - 1) / NP;
    const int o0 = part * opp, o1 = o0 + opp < na2 ? o0 + opp : na2;
    const bool worker = part < NP;
    // alpha words: byte offset of the source row (row na = zeros when the operator does not apply) | sign << 31
    if (!tabs_g) sec_build_table(s.unrank_a, s.rank_a, na, a, true, tmp, 512);
    __syncthreads();
    for (int i = tid; i < na2 * na; i += 512) {
        const uint32_t e = tabs_g ? tabs_g[i] : tmp[i];
        const uint32_t row = (e & 2048u) ? (e & 2047u) : (uint32_t)na;
        tA[i] = row * (uint32_t)(NBp * sizeof(double)) | ((e & 4096u) << 19);
    }
    __syncthreads();
    // beta words of this thread's beta string, once: source column (column nb = zero) and sign mask
    if (!tabs_g) sec_build_table(s.unrank_b, s.rank_b, nb, a, false, tmp, 512);
    __syncthreads();
    const uint16_t* tbb = tabs_g ? tabs_g + (size_t)na * na2 : tmp;
    uint32_t offB[MAXOP], sgnB[MAXOP];
#pragma unroll
    for (int u = 0; u < MAXOP; ++u) {
        const uint32_t e = (worker && o0 + u < o1 && ib < nb) ? tbb[(o0 + u) * nb + ib] : 0u;
        offB[u] = (e & 2048u) ? (e & 2047u) : (uint32_t)nb;
        sgnB[u] = (e & 4096u) << 19;
    }
    // the determinant behind column 4 kk + lq of a chunk, for the k-steps of this wave (<psi| V> on the vector ALUs)
    constexpr int MAXK = (SEC_RCH / 4 + 7) / 8;
    int poff[MAXK];
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int cc = 4 * (wave + 8 * i) + lq;
        poff[i] = cc < CH ? (cc / NBp) * NBp + (cc % NBp) : 0;      // (= cc: rows of a chunk are NBp apart in P too)
    }
    d4 acc[NT][NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
    double gpart[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) gpart[nt] = 0.0;
    __syncthreads();                                     // (tmp is Vc: all table words are out of it)
    for (int ch = split; ch < nchunk; ch += nsplit) {
        const int ia = ch * NRC + ial;
        if (worker && l < CH && !(probe == 1 && ch != split)) {
            if (ib < nb && ia < na) {
                const char* Pb = reinterpret_cast<const char*>(P) + (size_t)ib * sizeof(double);
                const double* rowb = P + (size_t)ia * NBp;
                const uint32_t* ta = tA + ia;
#pragma unroll
                for (int u0 = 0; u0 < MAXOP; u0 += 8) {
                    uint32_t wa[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) wa[u] = ta[(o0 + u0 + u < o1 ? o0 + u0 + u : o0) * na];
                    double va[8], vb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        va[u] = *reinterpret_cast<const double*>(Pb + (wa[u] & 0x7fffffffu));
                        vb[u] = rowb[offB[u0 + u]];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned long long xa = __builtin_bit_cast(unsigned long long, va[u]) ^
                                                      ((unsigned long long)(wa[u] & 0x80000000u) << 32);
                        const unsigned long long xb = __builtin_bit_cast(unsigned long long, vb[u]) ^
                                                      ((unsigned long long)sgnB[u0 + u] << 32);
                        if (o0 + u0 + u < o1)
                            Vc[(o0 + u0 + u) * SEC_RCHP + l] =
                                __builtin_bit_cast(double, xa) + __builtin_bit_cast(double, xb);
                    }
                }
            } else {
                for (int o = o0; o < o1; ++o) Vc[o * SEC_RCHP + l] = 0.0;
            }
        }
        __syncthreads();
        if (probe != 2) {
            const double* Pc = P + (size_t)ch * NRC * NBp;
#pragma unroll
            for (int i = 0; i < MAXK; ++i) {
                const int kk = wave + 8 * i;
                if (kk < KSC) {
                    const int cc = 4 * kk + lq;
                    double bv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = Vc[(nt * 16 + lr) * SEC_RCHP + cc];
                    const double pv = Pc[poff[i]];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) gpart[nt] += pv * bv[nt];
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) acc[mt][nt] = mfma_f64(bv[mt], bv[nt], acc[mt][nt]);
                }
            }
        }
        __syncthreads();
    }
    // the 8 waves' partial tiles are summed through LDS in fixed order, tile by tile (as sector_rdm_fused_kernel)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][(lq + 4 * i) * 16 + lr] = acc[mt][nt][i];
            __syncthreads();
            if (tid < 256) {
                double v = red[0][tid];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += red[w][tid];
                const int mrow = mt * 16 + tid / 16, col = nt * 16 + (tid & 15);
                const int row = (mrow % a) * a + mrow / a;            // product row (q,p) -> Gram row (p,q)
                double* Rb = R + ((size_t)split * batch + b) * (MTR * 16) * (NT * 16);
                Rb[(size_t)row * (NT * 16) + col] = v;
                if (mt != nt) Rb[(size_t)((col % a) * a + col / a) * (NT * 16) + mrow] = v;
            }
            __syncthreads();
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        red[wave][lq * 16 + lr] = gpart[nt];
        __syncthreads();
        if (tid < 16) {
            double v = 0.0;
            for (int w = 0; w < 8; ++w)
                for (int gq = 0; gq < 4; ++gq) v += red[w][gq * 16 + tid];
            R[(((size_t)split * batch + b) * (MTR * 16) + na2) * (NT * 16) + nt * 16 + tid] = v;
        }
        __syncthreads();
    }
}

// W[b][j][c] = sum_k Ms[k][j] (E_k psi_b)[c] without V in memory.  Wave w forms the 16 x 16 tiles of W rows
// 16 (w % NT) .. for the c-tiles (w / NT), (w / NT) + 8 / NT, ... of every chunk, its Ms fragments in registers.
template <int NT>
__global__ __launch_bounds__(512)
void sector_w_fused_kernel(const double* __restrict__ psi_c, const double* __restrict__ Ms, Sector s,
                           double* __restrict__ W, uint16_t* __restrict__ tab_out, const uint16_t* __restrict__ tabs_g)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb, a = s.ncas, na2 = a * a;
    double* src = lds;
    double* Vc = src + Dc + (Dc & 1);
    constexpr int SEC_CHP = SEC_CHP_W;
    uint16_t* tabA = reinterpret_cast<uint16_t*>(Vc + (size_t)na2 * SEC_CHP_MAX);
    uint16_t* tabB = tabA + (size_t)s.na * na2;
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    for (int i = tid; i < Dc; i += 512) src[i] = psi_c[b * Dc + i];
    sec_tables(s, tabs_g, tabA, tabB, 512);
    if (tab_out && blockIdx.x == 0 && blockIdx.y == 0) {
        // the excitation tables for the lambda kernel that follows this launch ([a^2][na] | [a^2][nb])
        __syncthreads();
        for (int i = tid; i < (s.na + s.nb) * na2; i += 512) tab_out[i] = tabA[i];
    }
    constexpr int KS = NT * 4;                           // k-steps over the a^2 = 16 NT rows of V
    const int jt = wave % NT, ct0 = wave / NT;           // this wave's W row tile, its first c-tile
    constexpr int CSTEP = 8 / NT;                        // c-tiles a wave skips (8 waves, NT row tiles)
    double af[KS];                                       // A[m = j][k] = Ms[k][16 jt + lr], k = 4 ks + lq
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = Ms[(size_t)(4 * ks + lq) * na2 + 16 * jt + lr];
    double* Wb = W + b * (size_t)na2 * Dc;
    const int nchunk = (Dc + SEC_CH - 1) / SEC_CH;
    __syncthreads();
    for (int ch = split; ch < nchunk; ch += nsplit) {
        const int c0 = ch * SEC_CH;
        sec_build_chunk<SEC_CHP>(src, s.na, s.nb, a, tabA, tabB, c0, Vc);
        __syncthreads();
        for (int ct = ct0; ct < SEC_CH / 16; ct += CSTEP) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = mfma_f64(af[ks], Vc[(4 * ks + lq) * SEC_CHP + 16 * ct + lr], acc);
            const int c = c0 + 16 * ct + lr;
            if (c < Dc)
#pragma unroll
                for (int i = 0; i < 4; ++i) Wb[(size_t)(16 * jt + lq + 4 * i) * Dc + c] = acc[i];
        }
        __syncthreads();
    }
}

// gamma[rs] = R[a^2][rs];  Gamma[pq,rs] = R[pq][rs] - delta_qr gamma[ps]
__global__ void sector_rdm_finish_kernel(const double* __restrict__ R, int ncas, int batch, int nsplit,
                                         double* __restrict__ gamma, double* __restrict__ Gamma)
{
    const int na2 = ncas * ncas, nrow = na2 + 1;
    const int MT = (nrow + 15) / 16, NT = (na2 + 15) / 16, ldr = NT * 16;
    const size_t b = blockIdx.y;
    const size_t tile_sz = (size_t)(MT * 16) * ldr;
    auto Rsum = [&](int off) -> double {
        double v = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) v += R[((size_t)sp * batch + b) * tile_sz + off];
        return v;
    };
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < na2 * na2 + na2;
         idx += gridDim.x * blockDim.x) {
        if (idx < na2) {
            gamma[b * na2 + idx] = Rsum(na2 * ldr + idx);
        } else {
            const int rem = idx - na2;
            const int pq = rem / na2, rs = rem - pq * na2;
            const int p = pq / ncas, q = pq - p * ncas, r = rs / ncas, s2 = rs - r * ncas;
            double g = Rsum(pq * ldr + rs);
            if (q == r) g -= Rsum(na2 * ldr + p * ncas + s2);
            Gamma[b * (size_t)na2 * na2 + rem] = g;
        }
    }
}

// ---- lambda = (Hop + Hop^T) psi -------------------------------------------------------------------
// E = psi^T Hop psi with Hop = sum c1e_pq E_pq + sum c2_pqrs E_pq E_rs, c1e[ps] = c1[ps] - sum_q c2[p,q,q,s].
// (E_pq E_rs)^T = E_sr E_qp, so  Hop + Hop^T = sum (c1e_pq + c1e_qp) E_pq + sum (c2[pq,rs] + c2[sr,qp]) E_pq E_rs
// and  lambda = sum_pq E_pq W_pq,  W_pq = sum_rs Ms[rs][pq] V_rs  with ONE symmetrised coefficient matrix
// (round 2 carried the transposed half as a second set of a^2 vectors W': twice the bytes written and read).
__global__ void sector_coeff_kernel(const double* __restrict__ c1, const double* __restrict__ c2,
                                    int ncas, const uint32_t* __restrict__ unrank_a,
                                    const uint32_t* __restrict__ unrank_b, double* __restrict__ Ms,
                                    long c1_bs, long c2_bs, long ms_bs)
{
    // blockIdx.y: the coefficient set (one per geometry of a stack: c1 + y c1_bs, c2 + y c2_bs -> Ms + y ms_bs)
    c1 += (size_t)blockIdx.y * c1_bs;
    c2 += (size_t)blockIdx.y * c2_bs;
    Ms += (size_t)blockIdx.y * ms_bs;
    // Ms [a^2][a^2]: row k = (r,s) of V, column j = (p,q) of W.  The one-body term u = sum_k s_k V_k rides
    // along: on the sector sum_p E_pp = N (the electron number), so adding s_k / N to the columns (p,p) of
    // row k adds sum_p E_pp (s_k / N) V_k = s_k V_k to lambda = sum_j E_j W_j -- no extra row of W to write
    // and read.   s_k = c1e[k] + c1e[swap k]
    const int na2 = ncas * ncas;
    const int nel = __popc(unrank_a[0]) + __popc(unrank_b[0]);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < na2 * na2;
         idx += gridDim.x * blockDim.x) {
        const int k = idx / na2, j = idx - k * na2;
        const int kr = k / ncas, ks = k - kr * ncas;          // V row k = E_rs psi
        const int jp = j / ncas, jq = j - jp * ncas;          // W row j = (p,q)
        double m = c2[(size_t)j * na2 + k] + c2[(size_t)(ks * ncas + kr) * na2 + (jq * ncas + jp)];
        if (jp == jq && nel > 0) {
            auto c1e = [&](int p, int s2) {
                double v = c1[p * ncas + s2];
                for (int q = 0; q < ncas; ++q) v -= c2[((size_t)(p * ncas + q) * ncas + q) * ncas + s2];
                return v;
            };
            m += (c1e(kr, ks) + c1e(ks, kr)) / nel;
        }
        Ms[(size_t)k * na2 + j] = m;
    }
}

// grid: (ceil(Dc/64), batch); block = 64 determinants x 4 slices of the (p,q) loop, summed in LDS
// in fixed order; string tables staged in LDS.
__global__ __launch_bounds__(256)
void sector_lambda_kernel(const double* __restrict__ W, Sector s, double* __restrict__ lam)
{
    extern __shared__ double lds[];
    double* part = lds;                                         // [4][64]
    uint32_t* tb = reinterpret_cast<uint32_t*>(part + 256);
    const int ns = 1 << s.ncas;
    int32_t* ra = reinterpret_cast<int32_t*>(tb);
    int32_t* rb = ra + ns;
    for (int i = threadIdx.x; i < ns; i += 256) { ra[i] = s.rank_a[i]; rb[i] = s.rank_b[i]; }
    const Sector sg = s;
    s.rank_a = ra;
    s.rank_b = rb;
    __syncthreads();
    const int Dc = s.na * s.nb, n = 2 * s.ncas, a = s.ncas, na2 = a * a;
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const size_t b = blockIdx.y;
    double acc = 0.0;
    if (c < Dc) {
        const double* Wb = W + b * (size_t)na2 * Dc;                 // [a^2][Dc]
        const uint32_t x = sec_full(sg, c);
        for (int pq = slice; pq < na2; pq += 4) {
            const int p = pq / a, q = pq - p * a;
            acc += sec_epq(Wb + (size_t)pq * Dc, s, n, p, q, x, c);       // E_pq W_pq
        }
    }
    part[slice * 64 + cl] = acc;
    __syncthreads();
    if (slice == 0 && c < Dc)
        lam[b * Dc + c] = part[cl] + part[64 + cl] + part[128 + cl] + part[192 + cl];
}

// The same with the excitation tables of the fused kernels (global, L1-resident): per (determinant, operator)
// two 16-bit table reads and two gathers from the row W_pq instead of the bit arithmetic of sec_epq.
// grid: (ceil(Dc/64), batch); block = 64 determinants x 4 slices of the (p,q) loop, summed in fixed order.
__global__ __launch_bounds__(256)
void sector_lambda_tab_kernel(const double* __restrict__ W, const uint16_t* __restrict__ tabA,
                              const uint16_t* __restrict__ tabB, int na, int nb, int ncas,
                              double* __restrict__ lam)
{
    __shared__ double part[4][64];
    const int Dc = na * nb, na2 = ncas * ncas;
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const size_t b = blockIdx.y;
    double acc = 0.0;
    if (c < Dc) {
        const double* Wb = W + b * (size_t)na2 * Dc;
        const int ia = c / nb, ib = c - ia * nb;
        for (int pq0 = slice; pq0 < na2; pq0 += 16) {
            uint32_t ea[4], eb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pq = pq0 + 4 * u;
                ea[u] = pq < na2 ? tabA[pq * na + ia] : 0u;
                eb[u] = pq < na2 ? tabB[pq * nb + ib] : 0u;
            }
            double va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pq = pq0 + 4 * u < na2 ? pq0 + 4 * u : 0;
                const double* row = Wb + (size_t)pq * Dc;
                va[u] = row[__umul24(ea[u] & 2047u, (unsigned)nb) + ib];
                vb[u] = row[ia * nb + (eb[u] & 2047u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool sa = ((ea[u] >> 12) ^ (eb[u] >> 13)) & 1u, sb = ((eb[u] >> 12) ^ (ea[u] >> 13)) & 1u;
                if (ea[u] & 2048u) acc += sa ? -va[u] : va[u];
                if (eb[u] & 2048u) acc += sb ? -vb[u] : vb[u];
            }
        }
    }
    part[slice][cl] = acc;
    __syncthreads();
    if (slice == 0 && c < Dc) lam[b * Dc + c] = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
}

// ---- round 4: lambda without W in memory ---------------------------------------------------------------------
// lambda = sum_jk Ms[k][j] E_j E_k psi with E_j = A_j + B_j (alpha part, beta part).  In the determinant basis
// whose signs are those of "all alpha operators, then all beta operators" -- psi' = sigma psi with
//   sigma(ia, ib) = (-1)^(sum over the alpha electrons i of the number of beta electrons in orbitals j < i)
// (the Jordan-Wigner order interleaves the spins, spin orbital 2p = alpha_p; sigma is the sign of sorting one into
// the other) -- A_j acts on the alpha string only and B_j on the beta string only, each with its OWN parity bit
// (the cross parities of the tables are exactly sigma(ia, ib) sigma(src, ib)).  With Psi' the na x nb matrix of
// psi':
//   lambda' = G_a Psi' + Psi' G_b^T + sum_j A_j ( sum_k K[k][j] B_k Psi' ),     K = Ms + Ms^T,
//   G_a = sum_jk Ms[k][j] A_j A_k  (na x na),   G_b = sum_jk Ms[k][j] B_j B_k  (nb x nb)
// (the string-driven sigma build of determinant CI).  G_a, G_b depend on the coefficients only: one small
// kernel per call.  The mixed term is LOCAL in the beta string for its gather (A_j moves alpha electrons only):
// a workgroup forms, for a chunk of SEC_LCH / LA beta strings and ALL alpha strings, the 64 vectors B_k Psi' in
// LDS, multiplies by K on the matrix cores INTO THE SAME LDS, and gathers with A_j there.  W (2.5 MB per
// CAS(8e,8o) state, 642 MB written and gathered again per 256 states: 290 + 190 us) never exists.
constexpr int SEC_LCH = 144;                 // determinants of a column chunk (whole beta strings x padded na)
constexpr int SEC_LCHP = SEC_LCH;            // LDS pitch of a row of the chunk: 144 doubles, rows 32 banks apart
static_assert(SEC_LCHP % 32 == 16, "k-step rows lq, lq + 1 of an MFMA operand read must fall into different bank halves");

__host__ __device__ inline size_t sec_lambda_lds_bytes(int na, int nb, int ncas)
{
    const size_t na2 = (size_t)ncas * ncas;
    const size_t Dt = ((size_t)nb + 1) * ((na + 8) & ~7);      // Psi' transposed, rows padded to LA (>= 1 zero column), + a zero row
    return (Dt + na2 * SEC_LCHP + 4 * SEC_LCH) * sizeof(double);
}

// grid: na + nb + ceil(Dc / 256) workgroups of 256 threads.  Workgroup r < na: row r of G_a (thread j < a^2: the
// chains A_j A_k from string r, its own partial row in LDS, summed over j in fixed order); the next nb: rows of
// G_b; the rest: sigma of 256 determinants each.
__global__ __launch_bounds__(256)
void sector_gmat_kernel(const double* __restrict__ Ms, Sector s, double* __restrict__ Ga,
                        double* __restrict__ Gb, double* __restrict__ sigma, uint16_t* __restrict__ tabs,
                        const uint16_t* __restrict__ tabs_g, long ms_bs, long g_bs)
{
    // blockIdx.y: the coefficient set (Ms + y ms_bs -> Ga + y g_bs, Gb + y g_bs); sigma and the tables do not depend
    // on the coefficients: set 0 makes them
    Ms += (size_t)blockIdx.y * ms_bs;
    Ga += (size_t)blockIdx.y * g_bs;
    Gb += (size_t)blockIdx.y * g_bs;
    const bool first_set = blockIdx.y == 0;
    // tabs [a^2][na] | [a^2][nb]: the excitation tables, for the kernels of the launches that follow
    // (tabs_g: the per-circuit copy, when the caller holds one: copied instead of built)
    extern __shared__ double lds[];
    const int a = s.ncas, na2 = a * a, tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk >= s.na + s.nb) {
        const int c = (blk - s.na - s.nb) * 256 + tid;
        if (first_set && c < s.na * s.nb) {
            const int ia = c / s.nb, ib = c - ia * s.nb;
            const uint32_t sa = s.unrank_a[ia], sb = s.unrank_b[ib];
            uint32_t par = 0;
            for (int i = 1; i < a; ++i)
                if (sa & (1u << (a - 1 - i))) par ^= __popc(sb & sec_orb_mask(a, 0, i - 1)) & 1u;
            sigma[c] = par ? -1.0 : 1.0;
        }
        return;
    }
    const bool alpha = blk < s.na;
    const int nstr = alpha ? s.na : s.nb, row = alpha ? blk : blk - s.na;
    double* part = lds;                                                    // [a^2][nstr]
    uint16_t* tab = reinterpret_cast<uint16_t*>(part + (size_t)na2 * nstr);  // [a^2][nstr]
    if (tabs_g) {
        const uint16_t* src_t = tabs_g + (alpha ? 0 : (size_t)s.na * na2);
        for (int i = tid; i < na2 * nstr; i += 256) tab[i] = src_t[i];
    } else {
        sec_build_table(alpha ? s.unrank_a : s.unrank_b, alpha ? s.rank_a : s.rank_b, nstr, a, alpha, tab, 256);
    }
    for (int i = tid; i < na2 * nstr; i += 256) part[i] = 0.0;
    __syncthreads();
    if (row == 0 && first_set) {
        uint16_t* out = tabs + (alpha ? 0 : (size_t)s.na * na2);
        for (int i = tid; i < na2 * nstr; i += 256) out[i] = tab[i];
        if (!alpha) {
            // the beta words as sector_lambda_pipe_kernel consumes them: byte offset of the source ROW of the
            // transposed Psi' (row nb = zeros when the operator does not apply) | sign << 31
            uint32_t* out32 = reinterpret_cast<uint32_t*>(tabs + (((size_t)(s.na + s.nb) * na2 + 1) & ~(size_t)1));
            const uint32_t rowb = (uint32_t)(((s.na + 8) & ~7) * sizeof(double));
            for (int i = tid; i < na2 * nstr; i += 256) {
                const uint32_t e = tab[i];
                out32[i] = ((e & 2048u) ? (e & 2047u) : (uint32_t)s.nb) * rowb | ((e & 4096u) << 19);
            }
        }
    }
    if (tid < na2) {
        const int j = tid;
        const uint32_t e1 = tab[j * nstr + row];
        if (e1 & 2048u) {
            const int s1 = e1 & 2047u;
            double* pj = part + (size_t)j * nstr;
            for (int k = 0; k < na2; ++k) {
                const uint32_t e2 = tab[k * nstr + s1];
                if (!(e2 & 2048u)) continue;
                const double m = Ms[(size_t)k * na2 + j];
                pj[e2 & 2047u] += (((e1 ^ e2) >> 12) & 1u) ? -m : m;
            }
        }
    }
    __syncthreads();
    double* G = (alpha ? Ga : Gb) + (size_t)row * nstr;
    for (int t = tid; t < nstr; t += 256) {
        double v = 0.0;
        for (int j = 0; j < na2; ++j) v += part[(size_t)j * nstr + t];
        G[t] = v;
    }
}

// lam'[b] = G_a Psi' + Psi' G_b^T (still in the sigma basis): grid = batch, 512 threads; Psi' zero padded in LDS,
// the 16 x 16 output tiles dealt to the 8 waves, both products on the matrix cores.
__global__ __launch_bounds__(512)
void sector_lambda_dense_kernel(const double* __restrict__ psi_c, const double* __restrict__ Ga,
                                const double* __restrict__ Gb, const double* __restrict__ sigma, int na, int nb,
                                double* __restrict__ lam, long g_bs, int group)
{
    // state b takes the matrices of coefficient set b / group (g_bs = 0: one set for all)
    Ga += (size_t)(blockIdx.x / group) * g_bs;
    Gb += (size_t)(blockIdx.x / group) * g_bs;
    extern __shared__ double lds[];
    const int Dc = na * nb, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int ta = (na + 15) / 16, tb = (nb + 15) / 16;
    const int ka = (na + 3) / 4, kb = (nb + 3) / 4;
    const int RP = ta * 16 > 4 * ka ? ta * 16 : 4 * ka;              // rows of the padded Psi'
    const int PP = (tb * 16 > 4 * kb ? tb * 16 : 4 * kb) + 4;        // its pitch
    double* P = lds;
    double* Gal = P + (size_t)RP * PP;                               // [na][na]
    double* Gbl = Gal + (size_t)na * na;                             // [nb][nb]
    const size_t b = blockIdx.x;
    for (int i = tid; i < RP * PP; i += 512) {
        const int r = i / PP, c = i - r * PP;
        P[i] = (r < na && c < nb) ? sigma[r * nb + c] * psi_c[b * Dc + r * nb + c] : 0.0;
    }
    for (int i = tid; i < na * na; i += 512) Gal[i] = Ga[i];
    for (int i = tid; i < nb * nb; i += 512) Gbl[i] = Gb[i];
    __syncthreads();
    for (int tile = wave; tile < ta * tb; tile += 8) {
        const int ti = tile / tb, tj = tile - ti * tb;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const int ia = 16 * ti + lr, ib = 16 * tj + lr;
        const int iac = ia < na ? ia : na - 1, ibc = ib < nb ? ib : nb - 1;
        for (int ks = 0; ks < ka; ++ks) {          // sum_k G_a[ia][k] Psi'[k][ib]  (rows k >= na of Psi' are zero)
            const int k = 4 * ks + lq;
            const double av = ia < na ? Gal[iac * na + (k < na ? k : na - 1)] : 0.0;
            acc = mfma_f64(av, P[k * PP + 16 * tj + lr], acc);
        }
        for (int ks = 0; ks < kb; ++ks) {          // sum_k Psi'[ia][k] G_b[ib][k]  (columns k >= nb of Psi' are zero)
            const int k = 4 * ks + lq;
            const double bv = ib < nb ? Gbl[ibc * nb + (k < nb ? k : nb - 1)] : 0.0;
            acc = mfma_f64(P[(16 * ti + lr) * PP + k], bv, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * ti + lq + 4 * i, c = 16 * tj + lr;
            if (r < na && c < nb) lam[b * Dc + r * nb + c] = acc[i];
        }
    }
}

// The mixed term, and the way back to the Jordan-Wigner basis:  lam[b] = sigma (lam'[b] + sum_j A_j Y_j),
// Y_j = sum_k K[k][j] B_k Psi'.  grid (batch, nsplit), 512 threads; the chunks of a state dealt to the nsplit
// workgroups; lam' comes from sector_lambda_dense_kernel (the launch before).
template <int NT>
__global__ __launch_bounds__(512)
void sector_lambda_fused_kernel(const double* __restrict__ psi_c, const double* __restrict__ Ms,
                                const double* __restrict__ sigma, const uint16_t* __restrict__ tabs, Sector s,
                                double* __restrict__ lam, int probe, long ms_bs, int group)
{
    // probe (timing only, wrong results): 1 no chunk build after the first, 2 no MFMA, 3 no gather
    // state b takes the coefficients of set b / group (ms_bs = 0: one set for all)
    Ms += (size_t)(blockIdx.x / group) * ms_bs;
    extern __shared__ double lds[];
    const int na = s.na, nb = s.nb, Dc = na * nb, a = s.ncas, na2 = a * a;
    const int LA = (na + 8) & ~7;                        // columns of one beta string inside a chunk (>= 1 zero column)
    const int NBC = SEC_LCH / LA;                        // beta strings per chunk (>= 1: checked by the host)
    const int CH = NBC * LA, CT = (CH + 15) / 16;
    // Psi' TRANSPOSED, [nb][LA]: the lanes of a wave hold consecutive alpha strings of one beta string, so the
    // amplitude reads of the build are consecutive words (row-major, 70 doubles apart, they were 4-way bank
    // conflicts: the build was LDS-bandwidth bound)
    // An element of the build or of the gather is one LDS read, one XOR (the sign, bit 63) and the store / add:
    // the table words are byte offsets (a ZERO row of Psi'^T, a zero column of Y when the operator does not apply)
    // and sign masks -- the beta ones from the 32-bit table sector_gmat_kernel leaves in memory, read one chunk
    // ahead; the alpha ones made once from the thread's alpha string.
    double* src = lds;                                   // [nb + 1][LA], row nb = zeros
    double* Vc = src + (size_t)(nb + 1) * LA;            // [a^2][SEC_LCHP]: B_k Psi' of the chunk, then Y
    double* red = Vc + (size_t)na2 * SEC_LCHP;           // [4][SEC_LCH]
    const uint16_t* tabA = tabs;
    const uint32_t* tabB32 = reinterpret_cast<const uint32_t*>(tabs + (((size_t)(na + nb) * na2 + 1) & ~(size_t)1));
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    for (int i = tid; i < (nb + 1) * LA; i += 512) {
        const int ib2 = i / LA, ia = i - ib2 * LA;
        src[i] = (ia < na && ib2 < nb) ? sigma[ia * nb + ib2] * psi_c[b * Dc + ia * nb + ib2] : 0.0;
    }
    constexpr int KS = NT * 4;
    const int jt = wave % NT, ct0 = wave / NT;
    constexpr int CSTEP = 8 / NT;
    constexpr int MAXT = (SEC_LCH / 16 + CSTEP - 1) / CSTEP;
    constexpr int MAXOP = NT == 4 ? 24 : 8;              // >= operators per thread (a^2 / 3 rounded up), multiple of 8
    double af[KS];                                       // A[m = j][k] = K[k][16 jt + lr], k = 4 ks + lq
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + lq, j = 16 * jt + lr;
        af[ks] = Ms[(size_t)k * na2 + j] + Ms[(size_t)j * na2 + k];
    }
    // the chunk column of this thread: NP = 512 / (columns of a chunk) threads per column, each with a share of
    // the operator range (3 x 22 operators for the 144 columns of CAS(8e,8o))
    const int LW = CT * 16;
    const int NP = 512 / LW < 4 ? 512 / LW : 4;
    const int part = tid / LW, l = tid - part * LW;
    const int ibl = l / LA, ial = l - ibl * LA;
    const int opp = (na2 + NP - 1) / NP;                 // operators per part
    const int o0 = part * opp, o1 = (o0 + opp < na2 ? o0 + opp : na2);
    const bool worker = part < NP;
    const int nchunk = (nb + NBC - 1) / NBC;
    __syncthreads();
    // table words in registers: the gather's (they depend on the alpha string of the thread's column only: the same
    // for every chunk) and the build's, read one chunk ahead
    uint32_t goff[MAXOP], gsgn[MAXOP], eb[MAXOP];
    const bool colive = worker && l < CH && ial < na;
#pragma unroll
    for (int u = 0; u < MAXOP; ++u) {
        const int o = o0 + u < o1 ? o0 + u : o0;
        const uint32_t e = (colive && o0 + u < o1) ? tabA[o * na + ial] : 0u;
        // (columns ial >= na of a beta string's block are zeros in Y: V's are)
        goff[u] = (uint32_t)((o * SEC_LCHP + ibl * LA + ((e & 2048u) ? (int)(e & 2047u) : na)) * (int)sizeof(double));
        gsgn[u] = (e & 4096u) << 19;
    }
    const uint32_t zero_row = (uint32_t)(nb * LA * (int)sizeof(double));
    auto load_eb = [&](int chn) {
        const int ibn = chn * NBC + ibl;
        const bool ok = colive && chn < nchunk && ibn < nb;
#pragma unroll
        for (int u = 0; u < MAXOP; ++u) eb[u] = (ok && o0 + u < o1) ? tabB32[(o0 + u) * nb + ibn] : zero_row;
    };
    const char* colb = reinterpret_cast<const char*>(src) + (size_t)(colive ? ial : 0) * sizeof(double);
    load_eb(split);
    for (int ch = split; ch < nchunk; ch += nsplit) {
        const int ib = ch * NBC + ibl;
        const bool live = worker && l < CH && ial < na && ib < nb;
        // lam' of this thread's determinant: needed at the end of the trip, asked for now
        double lam_dense = 0.0, sig = 1.0;
        if (live && part == 0) {
            lam_dense = lam[b * Dc + (size_t)ial * nb + ib];
            sig = sigma[ial * nb + ib];
        }
        // 1. Vc[k][l] = (B_k Psi')[ial, ib] = own sign * Psi'[ial, src_b(k, ib)]  (columns that are not live: zero row)
        if (worker && !(probe == 1 && ch != split)) {
#pragma unroll
            for (int u0 = 0; u0 < MAXOP; u0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double*>(colb + (eb[u0 + u] & 0x7fffffffu));
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (o0 + u0 + u < o1) {
                        const unsigned long long x = __builtin_bit_cast(unsigned long long, v[u]) ^
                                                     ((unsigned long long)(eb[u0 + u] & 0x80000000u) << 32);
                        Vc[(o0 + u0 + u) * SEC_LCHP + l] = __builtin_bit_cast(double, x);
                    }
            }
        }
        load_eb(ch + nsplit);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the global load / store of lam stay in flight)
        // 2. Y[j][l] = sum_k K[k][j] Vc[k][l]: the tiles of this wave in registers, then over Vc
        d4 acc[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int ct = ct0 + t * CSTEP;
            acc[t] = d4{0.0, 0.0, 0.0, 0.0};
            if (ct < CT && probe != 2) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc[t] = mfma_f64(af[ks], Vc[(4 * ks + lq) * SEC_LCHP + 16 * ct + lr], acc[t]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the global load / store of lam stay in flight)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int ct = ct0 + t * CSTEP;
            if (ct < CT)
#pragma unroll
                for (int i = 0; i < 4; ++i) Vc[(16 * jt + lq + 4 * i) * SEC_LCHP + 16 * ct + lr] = acc[t][i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the global load / store of lam stay in flight)
        // 3. sum_j (A_j Y_j)[ial, ib] = sum_j own sign * Y[j][(ibl, src_a(j, ial))]
        double sum = 0.0;
        if (live && probe != 3) {
            const char* yb = reinterpret_cast<const char*>(Vc);
#pragma unroll
            for (int u0 = 0; u0 < MAXOP; u0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double*>(yb + goff[u0 + u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned long long x = __builtin_bit_cast(unsigned long long, v[u]) ^
                                                 ((unsigned long long)gsgn[u0 + u] << 32);
                    if (o0 + u0 + u < o1) sum += __builtin_bit_cast(double, x);
                }
            }
        }
        if (worker && l < SEC_LCH) red[part * SEC_LCH + l] = sum;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the global load / store of lam stay in flight)
        if (live && part == 0) {
            double t = red[l];
            for (int pp = 1; pp < NP; ++pp) t += red[pp * SEC_LCH + l];
            lam[b * Dc + (size_t)ial * nb + ib] = sig * (lam_dense + t);
        }
        // (the next chunk's build writes Vc and, three barriers later, red: both are free by then -- every
        // thread has passed the barrier above, and the last readers of Vc are in front of it)
    }
}

// The same mixed term with the phases of sector_lambda_fused_kernel OVERLAPPED (a^2 = 64): that kernel runs build,
// product and gather one after the other (146 + ~50 + ~30-55 us of its 311 at 256 CAS(8e,8o) states: one workgroup per
// CU, nothing else to fill the matrix cores while it builds).  Here waves 0-3 (one per SIMD) only multiply and waves
// 4-7 only build and gather, on chunks of ONE beta string (all alpha strings: 70 columns in 5 tiles):
//   interval 1 of trip t:  multipliers  Y(t) = K^T V(t) from Vbuf[t & 1] into registers
//                          helpers      gather of Y(t-1) from Ybuf into partial sums, then build of V(t+1) into
//                                       Vbuf[(t+1) & 1]
//   interval 2 of trip t:  multipliers  Y(t) registers -> Ybuf;   helpers: lam of chunk t-1 = sigma (dense + sums)
// two workgroup barriers per chunk.  MEASURED, NOT ADOPTED (option sector_lambda_w = 4 selects it): 344 us against 311.
// The cycle counters of a workgroup (sector_probe = 9) say why: the multiplier wave of a SIMD needs 5.9 K cycles per
// chunk for its 80 products and then WAITS 5 K cycles for the helper wave of the same SIMD, whose ~360 instructions
// (two LDS operations and ~4 integer operations per element, the table words of the next chunk) take 10.6 K cycles
// beside the products and 6 K alone: v_mfma_f64 runs on the SIMD's fp64 vector path, the older wave owns it, and
// the helper gets about one instruction in per product.  Splitting the roles over SIMDs instead (two SIMDs
// multiply, two help) would halve the matrix rate of the CU.  On gfx950 fp64 matrix work cannot be hidden behind
// vector work of the same workgroup; the phases of sector_lambda_fused_kernel stay one after the other.
// The excitation tables stay in memory (L1 / L2: every helper thread needs 22 + 22
// words per chunk, the build's read one chunk ahead), so that Psi'^T (40 KB), two V buffers (2 x 41 KB) and Y (37 KB)
// fit the LDS.
__device__ long long g_pipe_cyc[16];        // probe 9: cycles of workgroup (0,0): [0..3] multiplier wave 0, [4..7] helper wave 4
constexpr int SEC_PCOL = 80;                 // columns of a chunk buffer (70 alpha strings in 5 tiles), pitch of V
constexpr int SEC_PYP = 72;                  // pitch of Y

__host__ __device__ inline size_t sec_lambda_pipe_lds_bytes(int na, int nb, int ncas)
{
    const size_t na2 = (size_t)ncas * ncas, LA = ((size_t)na + 8) & ~(size_t)7;
    return (((size_t)nb + 1) * LA + 2 * na2 * SEC_PCOL + na2 * SEC_PYP + 3 * SEC_PCOL) * sizeof(double);
}

__global__ __launch_bounds__(512)
void sector_lambda_pipe_kernel(const double* __restrict__ psi_c, const double* __restrict__ Ms,
                               const double* __restrict__ sigma, const uint16_t* __restrict__ tabs, Sector s,
                               double* __restrict__ lam, int probe)
{
    // probe (timing only, wrong results): 1 no build / gather, 2 no MFMA
    extern __shared__ double lds[];
    constexpr int na2 = 64, KS = 16, CT = SEC_PCOL / 16;
    const int na = s.na, nb = s.nb, Dc = na * nb;
    const int LA = (na + 8) & ~7;
    double* src = lds;                                   // [nb + 1][LA] sigma psi, transposed; row nb = zeros
    double* Vb = src + (size_t)(nb + 1) * LA;            // [2][a^2][SEC_PCOL]
    double* Yb = Vb + 2 * na2 * SEC_PCOL;                // [a^2][SEC_PYP]; columns >= na are zeros (V's are)
    double* red = Yb + na2 * SEC_PYP;                    // [3][SEC_PCOL]
    const uint16_t* tabA = tabs;
    const uint32_t* tabB32 = reinterpret_cast<const uint32_t*>(tabs + (((size_t)(na + nb) * na2 + 1) & ~(size_t)1));
    const size_t b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    for (int i = tid; i < (nb + 1) * LA; i += 512) {
        const int ib2 = i / LA, ia = i - ib2 * LA;
        src[i] = (ia < na && ib2 < nb) ? sigma[ia * nb + ib2] * psi_c[b * Dc + ia * nb + ib2] : 0.0;
    }
    const bool mult = wave < 4;
    // multipliers: row tile jt = wave of Y, all CT column tiles
    double af[KS];
    if (mult) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 4 * ks + lq, j = 16 * wave + lr;
            af[ks] = Ms[(size_t)k * na2 + j] + Ms[(size_t)j * na2 + k];
        }
    }
    // helpers: column (alpha string) and operator share of the thread
    const int h = tid - 256;
    const int part = h >= 0 ? h / SEC_PCOL : 3, col = h >= 0 ? h - part * SEC_PCOL : 0;
    const bool helper = !mult && part < 3;
    constexpr int MAXOP = 24, OPP = 22;
    const int o0 = part * OPP, o1 = o0 + OPP < na2 ? o0 + OPP : na2;
    const bool colive = helper && col < na;
    // gather words, once: byte offset into Y of the source element (a zero column when the operator does not apply)
    // and the sign mask -- an element of the gather is one LDS read, one XOR, one add
    uint32_t goff[MAXOP], gsgn[MAXOP], eb[MAXOP];
#pragma unroll
    for (int u = 0; u < MAXOP; ++u) {
        const int o = o0 + u < o1 ? o0 + u : o0;
        const uint32_t e = (colive && o0 + u < o1) ? tabA[o * na + col] : 0u;
        goff[u] = (uint32_t)((o * SEC_PYP + ((e & 2048u) ? (int)(e & 2047u) : na)) * (int)sizeof(double));
        gsgn[u] = (e & 4096u) << 19;
    }
    const int nchunk = (nb - split + nsplit - 1) / nsplit;       // chunks (beta strings) of this workgroup
    const uint32_t zero_row = (uint32_t)(nb * LA * (int)sizeof(double));
    auto load_eb = [&](int t) {
        const int ibn = split + t * nsplit;
        const bool ok = colive && t < nchunk && ibn < nb;
#pragma unroll
        for (int u = 0; u < MAXOP; ++u) eb[u] = (ok && o0 + u < o1) ? tabB32[(o0 + u) * nb + ibn] : zero_row;
    };
    const char* colb = reinterpret_cast<const char*>(src) + (size_t)(col < LA ? col : 0) * sizeof(double);
    auto build = [&](int t) {           // V(t)[k][col] = own sign * Psi'[col, src_b(k, ib)]; eb holds chunk t's words
        double* V = Vb + (size_t)(t & 1) * na2 * SEC_PCOL;
#pragma unroll
        for (int u0 = 0; u0 < MAXOP; u0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double*>(colb + (eb[u0 + u] & 0x7fffffffu));
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (o0 + u0 + u < o1) {
                    const unsigned long long x = __builtin_bit_cast(unsigned long long, v[u]) ^
                                                 ((unsigned long long)(eb[u0 + u] & 0x80000000u) << 32);
                    V[(o0 + u0 + u) * SEC_PCOL + col] = __builtin_bit_cast(double, x);
                }
        }
    };
    __syncthreads();
    if (helper) {
        load_eb(0);
        build(0);
        load_eb(1);
    }
    __syncthreads();
    d4 acc[CT];
    double lam_dense = 0.0, sig = 1.0;
    long long cy[4] = {0, 0, 0, 0};
    const bool timed = probe == 9 && blockIdx.x == 0 && blockIdx.y == 0 && (wave == 0 || wave == 4);
    for (int t = 0; t <= nchunk; ++t) {
        long long c0 = timed ? (long long)__builtin_readcyclecounter() : 0;
        // ---- interval 1
        if (mult) {
            if (t < nchunk && probe != 2) {
                const double* V = Vb + (size_t)(t & 1) * na2 * SEC_PCOL + lq * SEC_PCOL + lr;
                // (one wave per SIMD multiplies: the operands of the next column tile are read before the products
                // of this one, or every tile would start with an exposed LDS round trip)
                double bo[2][KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bo[0][ks] = V[4 * ks * SEC_PCOL];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (ct + 1 < CT) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) bo[(ct + 1) & 1][ks] = V[4 * ks * SEC_PCOL + 16 * (ct + 1)];
                    }
                    acc[ct] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc[ct] = mfma_f64(af[ks], bo[ct & 1][ks], acc[ct]);
                }
            }
        } else if (helper && probe != 1) {
            if (t >= 1) {
                // gather of chunk t - 1: sum_j own sign * Y[j][src_a(j, col)]
                const int ibp = split + (t - 1) * nsplit;
                if (colive && part == 0) {          // (both needed in interval 2: asked for now)
                    lam_dense = lam[b * Dc + (size_t)col * nb + ibp];
                    sig = sigma[col * nb + ibp];
                }
                double sum = 0.0;
                const char* yb = reinterpret_cast<const char*>(Yb);
#pragma unroll
                for (int u0 = 0; u0 < MAXOP; u0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double*>(yb + goff[u0 + u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned long long x = __builtin_bit_cast(unsigned long long, v[u]) ^
                                                     ((unsigned long long)gsgn[u0 + u] << 32);
                        if (o0 + u0 + u < o1) sum += __builtin_bit_cast(double, x);
                    }
                }
                red[part * SEC_PCOL + col] = sum;
            }
            if (t + 1 < nchunk) {
                build(t + 1);
                load_eb(t + 2);
            }
        }
        long long c1 = timed ? (long long)__builtin_readcyclecounter() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        long long c2 = timed ? (long long)__builtin_readcyclecounter() : 0;
        // ---- interval 2
        if (mult) {
            if (t < nchunk) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    if (16 * ct + lr < SEC_PYP)
#pragma unroll
                        for (int i = 0; i < 4; ++i) Yb[(16 * wave + lq + 4 * i) * SEC_PYP + 16 * ct + lr] = acc[ct][i];
            }
        } else if (helper && part == 0 && colive && t >= 1) {
            const int ibp = split + (t - 1) * nsplit;
            const double tot = red[col] + red[SEC_PCOL + col] + red[2 * SEC_PCOL + col];
            lam[b * Dc + (size_t)col * nb + ibp] = sig * (lam_dense + tot);
        }
        long long c3 = timed ? (long long)__builtin_readcyclecounter() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (timed) {
            const long long c4 = (long long)__builtin_readcyclecounter();
            cy[0] += c1 - c0; cy[1] += c2 - c1; cy[2] += c3 - c2; cy[3] += c4 - c3;
        }
    }
    if (timed && lane == 0)
        for (int i = 0; i < 4; ++i) g_pipe_cyc[(wave == 0 ? 0 : 4) + i] = cy[i];
}


// ---- adjoint sweep: grid = batch --------------------------------------------------------------------
template <int MAXIT>
__global__ __launch_bounds__(SEC_THREADS)
void sector_adjoint_kernel(const double* __restrict__ theta, int n_theta,
                           const oovqe_gate_t* __restrict__ gates, int n_gates, Sector s,
                           const double* __restrict__ psi_c, const double* __restrict__ lam,
                           double* __restrict__ dtheta)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb;
    double* ps = lds;                                   // [Dc]
    double* lm = ps + Dc;                               // [Dc]
    double* cs = lm + Dc;                               // [n_gates][2]
    double* part = cs + 2 * n_gates;                    // [n_gates][16 waves]
    oovqe_gate_t* gl = reinterpret_cast<oovqe_gate_t*>(part + (size_t)n_gates * (SEC_THREADS / 64) + n_theta);
    uint32_t* xfull;
    s = sec_stage_lds(s, reinterpret_cast<uint32_t*>(gl + n_gates), &xfull, SEC_THREADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const double* th = theta + (size_t)b * n_theta;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gates);
        uint32_t* dst = reinterpret_cast<uint32_t*>(gl);
        for (int i = tid; i < n_gates * (int)(sizeof(oovqe_gate_t) / 4); i += SEC_THREADS) dst[i] = src[i];
        for (int g = tid; g < n_gates; g += SEC_THREADS) {
            const int ti = gates[g].theta_idx;
            double sn = 0.0, c = 1.0;
            if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
            cs[2 * g] = c;
            cs[2 * g + 1] = sn;
        }
        for (int d = tid; d < Dc; d += SEC_THREADS) {
            ps[d] = psi_c[(size_t)b * Dc + d];
            lm[d] = lam[(size_t)b * Dc + d];
        }
    }
    __syncthreads();
    for (int g = n_gates - 1; g >= 0; --g) {
        const oovqe_gate_t gt = gl[g];
        if (gt.theta_idx < 0) continue;
        const uint32_t fm = gt.mask_hi | gt.mask_lo;
        const double c = cs[2 * g], sn = cs[2 * g + 1];
        double acc = 0.0;
        int e[MAXIT];
        double pi[MAXIT];
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int d = tid + i * SEC_THREADS;
            e[i] = -1;
            if (d < Dc) {
                const uint32_t x = xfull[d];
                if ((x & fm) == gt.mask_hi) {
                    e[i] = sec_rank(s, x ^ fm);
                    pi[i] = (__popc(x & gt.mask_par) & 1) ? -1.0 : 1.0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            if (e[i] >= 0) {
                const int d = tid + i * SEC_THREADS, ee = e[i];
                const double px = ps[d], py = ps[ee], lx = lm[d], ly = lm[ee];
                acc += pi[i] * (lx * py - ly * px);          // lambda^T A psi on this pair
                const double ps_ = pi[i] * sn;
                ps[d] = c * px - ps_ * py;                   // U^T
                ps[ee] = c * py + ps_ * px;
                lm[d] = c * lx - ps_ * ly;
                lm[ee] = c * ly + ps_ * lx;
            }
        }
        // deterministic reduction: wave shuffle tree now, the 16 wave partials of every gate are
        // summed in fixed order after the sweep (one barrier per gate instead of two and no
        // single-thread section inside the sweep)
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) part[(size_t)g * (SEC_THREADS / 64) + wave] = acc;
        __syncthreads();
    }
    // gth[k] = sum over the gates driven by theta_k, in the order of the sweep (last gate first)
    for (int k = tid; k < n_theta; k += SEC_THREADS) {
        double acc = 0.0;
        for (int g = n_gates - 1; g >= 0; --g) {
            if (gl[g].theta_idx != k) continue;
            double tot = 0.0;
            for (int w = 0; w < SEC_THREADS / 64; ++w) tot += part[(size_t)g * (SEC_THREADS / 64) + w];
            acc += 0.5 * (double)gl[g].sign * tot;
        }
        dtheta[(size_t)b * n_theta + k] = acc;
    }
}

// ---- round 4: the gates as PAIR LISTS -----------------------------------------------------------------------------
// A Givens pass rotates the pairs (d, e = rank(x_d ^ flip)) of determinants with (x_d & flip) == mask_hi: 400 of the
// 4 900 determinants of CAS(8e,8o) for a pair double excitation.  The sweeps above find them per gate and per
// workgroup -- every thread reads the full index of its determinants, tests the mask, compacts two bit strings and
// looks two ranks up: 1.4 us per gate of dependent LDS round trips -- although they depend on the gate table alone.
// sector_pairs_kernel lists them once per circuit (one 32-bit word per pair: d | e << 15 | parity << 31, ascending
// d); the sweeps below read the words of the NEXT gate from L2 while they rotate the pairs of the current one.
constexpr int SEC_PL_THREADS = 512;

__host__ __device__ inline int sec_pair_stride(int Dc) { return ((Dc / 2 + 63) / 64) * 64 + 64; }

// grid = n_gates + 2, 1024 threads.  pairs: [n_gates] counts | [n_gates][stride] words | the excitation tables.
__global__ __launch_bounds__(1024)
void sector_pairs_kernel(const oovqe_gate_t* __restrict__ gates, int n_gates, Sector s, uint32_t* __restrict__ pairs)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Dc = s.na * s.nb, stride = sec_pair_stride(Dc);
    if (g >= n_gates) {
        // the two excitation tables of the sector ([a^2][na] | [a^2][nb], 16-bit), behind the lists
        uint16_t* tabs = reinterpret_cast<uint16_t*>(pairs + (size_t)n_gates * (1 + stride));
        if (g == n_gates) sec_build_table(s.unrank_a, s.rank_a, s.na, s.ncas, true, tabs, 1024);
        else sec_build_table(s.unrank_b, s.rank_b, s.nb, s.ncas, false, tabs + (size_t)s.na * s.ncas * s.ncas, 1024);
        return;
    }
    const oovqe_gate_t gt = gates[g];
    const uint32_t fm = gt.mask_hi | gt.mask_lo;
    uint32_t* out = pairs + n_gates + (size_t)g * stride;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int d0 = 0; d0 < Dc; d0 += 1024) {
        const int d = d0 + tid;
        bool hit = false;
        uint32_t word = 0;
        if (d < Dc && gt.theta_idx >= 0) {
            const uint32_t x = sec_full(s, d);
            if ((x & fm) == gt.mask_hi) {
                hit = true;
                word = (uint32_t)d | ((uint32_t)sec_rank(s, x ^ fm) << 15) | ((uint32_t)(__popc(x & gt.mask_par) & 1) << 31);
            }
        }
        const unsigned long long m = __ballot(hit);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (hit) out[off + before] = word;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) pairs[g] = (uint32_t)base;
}

// forward circuit from the pair lists: grid = batch, 512 threads; LDS: the vector, cos / sin, counts
template <int MAXP>
__global__ __launch_bounds__(SEC_PL_THREADS)
void sector_circuit_pl_kernel(const double* __restrict__ theta, int n_theta, const oovqe_gate_t* __restrict__ gates,
                              int n_gates, Sector s, uint32_t init_index, const uint32_t* __restrict__ pairs,
                              double* __restrict__ psi_c)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb, c0 = sec_rank(s, init_index);
    double* st = lds;                                   // [Dc]
    double* cs = st + Dc;                               // [n_gates][2]
    int* cnt = reinterpret_cast<int*>(cs + 2 * n_gates);   // [n_gates], -1: not a rotation
    const int tid = threadIdx.x, b = blockIdx.x;
    const int stride = sec_pair_stride(Dc);
    const uint32_t* lists = pairs + n_gates;
    const double* th = theta + (size_t)b * n_theta;
    for (int g = tid; g < n_gates; g += SEC_PL_THREADS) {
        const int ti = gates[g].theta_idx;
        double sn = 0.0, c = 1.0;
        if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
        cs[2 * g] = c;
        cs[2 * g + 1] = sn;
        cnt[g] = ti >= 0 ? (int)pairs[g] : -1;
    }
    for (int d = tid; d < Dc; d += SEC_PL_THREADS) st[d] = (d == c0) ? 1.0 : 0.0;
    __syncthreads();
    uint32_t w[MAXP], nw[MAXP];
    auto fetch = [&](int g, uint32_t (&dst)[MAXP]) {
        const int n = g < n_gates ? cnt[g] : 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int idx = tid + i * SEC_PL_THREADS;
            dst[i] = idx < n ? lists[(size_t)g * stride + idx] : 0u;
        }
    };
    fetch(0, w);
    for (int g = 0; g < n_gates; ++g) {
        fetch(g + 1, nw);                               // (in flight while this gate's pairs rotate)
        const int n = cnt[g];
        if (n > 0) {
            const double c = cs[2 * g], sn = cs[2 * g + 1];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (tid + i * SEC_PL_THREADS < n) {
                    const int d = w[i] & 0x7fffu, e = (w[i] >> 15) & 0x7fffu;
                    const double pi = (w[i] >> 31) ? -sn : sn;
                    const double ax = st[d], ay = st[e];
                    st[d] = c * ax + pi * ay;
                    st[e] = c * ay - pi * ax;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the next words stay in flight)
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) w[i] = nw[i];
    }
    __syncthreads();
    for (int d = tid; d < Dc; d += SEC_PL_THREADS) psi_c[(size_t)b * Dc + d] = st[d];
}

// sector_circuit_deriv_kernel from the pair lists: grid (batch, n_out).  A differentiated gate also annihilates
// every amplitude it leaves alone: its pairs' new values are formed in registers, the vector is cleared, the values
// are written (two more barriers, for at most two gates of an output).
template <int MAXP>
__global__ __launch_bounds__(SEC_PL_THREADS)
void sector_circuit_deriv_pl_kernel(const double* __restrict__ theta, int n_theta,
                                    const oovqe_gate_t* __restrict__ gates, int n_gates, Sector s,
                                    uint32_t init_index, const uint32_t* __restrict__ pairs,
                                    const int32_t* __restrict__ deriv, int n_out, double* __restrict__ psi_c)
{
    extern __shared__ double lds[];
    const int Dc = s.na * s.nb, c0 = sec_rank(s, init_index);
    double* st = lds;                                   // [Dc]
    double* cs = st + Dc;                               // [n_gates][2]
    int* cnt = reinterpret_cast<int*>(cs + 2 * n_gates);   // [n_gates], -1: not a rotation
    const int tid = threadIdx.x, b = blockIdx.x, o = blockIdx.y;
    const int stride = sec_pair_stride(Dc);
    const uint32_t* lists = pairs + n_gates;
    const double* th = theta + (size_t)b * n_theta;
    const int ga = deriv[2 * o], gb = deriv[2 * o + 1];
    for (int g = tid; g < n_gates; g += SEC_PL_THREADS) {
        const int ti = gates[g].theta_idx;
        double sn = 0.0, c = 1.0;
        if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
        const int order = (g == ga ? 1 : 0) + (g == gb ? 1 : 0);
        const double h = 0.5 * (double)gates[g].sign;
        if (order == 1) { const double c1 = -h * sn, s1 = h * c; c = c1; sn = s1; }
        if (order == 2) { c *= -0.25; sn *= -0.25; }
        cs[2 * g] = c;
        cs[2 * g + 1] = sn;
        cnt[g] = ti >= 0 ? (int)pairs[g] : -1;
    }
    for (int d = tid; d < Dc; d += SEC_PL_THREADS) st[d] = (d == c0) ? 1.0 : 0.0;
    __syncthreads();
    uint32_t w[MAXP], nw[MAXP];
    auto fetch = [&](int g, uint32_t (&dst)[MAXP]) {
        const int n = g < n_gates ? cnt[g] : 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int idx = tid + i * SEC_PL_THREADS;
            dst[i] = idx < n ? lists[(size_t)g * stride + idx] : 0u;
        }
    };
    fetch(0, w);
    for (int g = 0; g < n_gates; ++g) {
        fetch(g + 1, nw);
        const int n = cnt[g];
        if (n >= 0) {
            const double c = cs[2 * g], sn = cs[2 * g + 1];
            const bool differentiated = g == ga || g == gb;
            double vd[MAXP], ve[MAXP];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (tid + i * SEC_PL_THREADS < n) {
                    const int d = w[i] & 0x7fffu, e = (w[i] >> 15) & 0x7fffu;
                    const double pi = (w[i] >> 31) ? -sn : sn;
                    const double ax = st[d], ay = st[e];
                    vd[i] = c * ax + pi * ay;
                    ve[i] = c * ay - pi * ax;
                    if (!differentiated) { st[d] = vd[i]; st[e] = ve[i]; }
                }
            }
            if (differentiated) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int d = tid; d < Dc; d += SEC_PL_THREADS) st[d] = 0.0;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    if (tid + i * SEC_PL_THREADS < n) {
                        st[w[i] & 0x7fffu] = vd[i];
                        st[(w[i] >> 15) & 0x7fffu] = ve[i];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) w[i] = nw[i];
    }
    __syncthreads();
    for (int d = tid; d < Dc; d += SEC_PL_THREADS) psi_c[((size_t)b * n_out + o) * Dc + d] = st[d];
}

// reverse sweep from the pair lists (the arithmetic of sector_adjoint_kernel on every pair)
template <int MAXP>
__global__ __launch_bounds__(SEC_PL_THREADS)
void sector_adjoint_pl_kernel(const double* __restrict__ theta, int n_theta, const oovqe_gate_t* __restrict__ gates,
                              int n_gates, int Dc, const uint32_t* __restrict__ pairs,
                              const double* __restrict__ psi_c, const double* __restrict__ lam,
                              double* __restrict__ dtheta)
{
    extern __shared__ double lds[];
    constexpr int NWV = SEC_PL_THREADS / 64;
    double* ps = lds;                                   // [Dc]
    double* lm = ps + Dc;                               // [Dc]
    double* cs = lm + Dc;                               // [n_gates][2]
    double* part = cs + 2 * n_gates;                    // [n_gates][NWV]
    int* cnt = reinterpret_cast<int*>(part + (size_t)n_gates * NWV);   // [n_gates]
    int* tix = cnt + n_gates;                           // [n_gates] theta index
    int* sgn = tix + n_gates;                           // [n_gates] sign of the angle
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int stride = sec_pair_stride(Dc);
    const uint32_t* lists = pairs + n_gates;
    const double* th = theta + (size_t)b * n_theta;
    for (int g = tid; g < n_gates; g += SEC_PL_THREADS) {
        const int ti = gates[g].theta_idx;
        double sn = 0.0, c = 1.0;
        if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &c);
        cs[2 * g] = c;
        cs[2 * g + 1] = sn;
        cnt[g] = ti >= 0 ? (int)pairs[g] : -1;
        tix[g] = ti;
        sgn[g] = gates[g].sign;
    }
    for (int i = tid; i < n_gates * NWV; i += SEC_PL_THREADS) part[i] = 0.0;
    for (int d = tid; d < Dc; d += SEC_PL_THREADS) {
        ps[d] = psi_c[(size_t)b * Dc + d];
        lm[d] = lam[(size_t)b * Dc + d];
    }
    __syncthreads();
    uint32_t w[MAXP], nw[MAXP];
    auto fetch = [&](int g, uint32_t (&dst)[MAXP]) {
        const int n = g >= 0 ? cnt[g] : 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int idx = tid + i * SEC_PL_THREADS;
            dst[i] = idx < n ? lists[(size_t)g * stride + idx] : 0u;
        }
    };
    fetch(n_gates - 1, w);
    for (int g = n_gates - 1; g >= 0; --g) {
        fetch(g - 1, nw);
        const int n = cnt[g];
        if (n > 0) {
            const double c = cs[2 * g], sn = cs[2 * g + 1];
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (tid + i * SEC_PL_THREADS < n) {
                    const int d = w[i] & 0x7fffu, ee = (w[i] >> 15) & 0x7fffu;
                    const double pi = (w[i] >> 31) ? -1.0 : 1.0;
                    const double px = ps[d], py = ps[ee], lx = lm[d], ly = lm[ee];
                    acc += pi * (lx * py - ly * px);              // lambda^T A psi on this pair
                    const double ps_ = pi * sn;
                    ps[d] = c * px - ps_ * py;                    // U^T
                    ps[ee] = c * py + ps_ * px;
                    lm[d] = c * lx - ps_ * ly;
                    lm[ee] = c * ly + ps_ * lx;
                }
            }
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0) part[(size_t)g * NWV + wave] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) w[i] = nw[i];
    }
    __syncthreads();
    // gth[k] = sum over the gates driven by theta_k, in the order of the sweep (last gate first)
    for (int k = tid; k < n_theta; k += SEC_PL_THREADS) {
        double acc = 0.0;
        for (int g = n_gates - 1; g >= 0; --g) {
            if (tix[g] != k) continue;
            double tot = 0.0;
            for (int wv = 0; wv < NWV; ++wv) tot += part[(size_t)g * NWV + wv];
            acc += 0.5 * (double)sgn[g] * tot;
        }
        dtheta[(size_t)b * n_theta + k] = acc;
    }
}

Sector make_sector(const uint32_t* ua, const uint32_t* ub, const int32_t* ra, const int32_t* rb, int na,
                   int nb, int ncas)
{
    Sector s;
    s.unrank_a = ua; s.unrank_b = ub; s.rank_a = ra; s.rank_b = rb;
    s.na = na; s.nb = nb; s.ncas = ncas;
    return s;
}

size_t circuit_lds(int na, int nb, int ncas, int n_gates)
{
    return ((size_t)na * nb + 2 * n_gates) * sizeof(double) + (size_t)n_gates * sizeof(oovqe_gate_t) +
           sec_lds_words(na, nb, ncas) * 4;
}

size_t adjoint_lds(int na, int nb, int ncas, int n_gates, int n_theta)
{
    return ((size_t)2 * na * nb + 2 * n_gates + (size_t)n_gates * (SEC_THREADS / 64) + n_theta) *
               sizeof(double) +
           (size_t)n_gates * sizeof(oovqe_gate_t) + sec_lds_words(na, nb, ncas) * 4;
}

}  // namespace

// (measurement hook of sector_lambda_pipe_kernel, option sector_probe = 9; not part of include/oovqe.h)
extern "C" int oovqe_sector_pipe_cycles(long long* out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pipe_cyc), 16 * sizeof(long long)) == hipSuccess ? 0 : OOVQE_ERR_HIP;
}

static int sector_cu_count()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

extern "C" int oovqe_sector_state_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                     int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                     const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                     int na, int nb, int batch, const uint32_t* pairs, int max_pairs,
                                     double* psi_c, double* psi_dense, oovqe_stream_t stream);

extern "C" int64_t oovqe_sector_pairs_size(int n_gates, int na, int nb)
{
    if (n_gates < 1 || na < 1 || nb < 1) return 0;
    return (int64_t)n_gates * (1 + sec_pair_stride(na * nb));      // 32-bit words (the tables: oovqe_sector_tables_size)
}

// 32-bit words of the sector's excitation tables ([a^2][na] | [a^2][nb], 16-bit entries) that oovqe_sector_pairs
// writes behind the lists: its buffer holds oovqe_sector_pairs_size + oovqe_sector_tables_size words, the tables
// start at word oovqe_sector_pairs_size.
extern "C" int64_t oovqe_sector_tables_size(int ncas, int na, int nb)
{
    if (ncas < 1 || na < 1 || nb < 1) return 0;
    return ((int64_t)(na + nb) * ncas * ncas + 1) / 2;
}

// The pair lists of a gate table (once per circuit): pairs [oovqe_sector_pairs_size] 32-bit words -- per gate the
// number of pairs, then per gate its pairs d | e << 15 | parity << 31 in ascending d.
extern "C" int oovqe_sector_pairs(const oovqe_gate_t* gates, int n_gates, int ncas, const uint32_t* unrank_a,
                                  const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b, int na,
                                  int nb, uint32_t* pairs, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(gates && unrank_a && unrank_b && rank_a && rank_b && pairs, "sector_pairs: null pointer");
    OOVQE_REQUIRE(ncas >= 1 && ncas <= 13 && na >= 1 && nb >= 1 && n_gates >= 1 && n_gates <= 65535 &&
                  (long)na * nb <= 32767, "sector_pairs: bad sizes (at most 32 767 determinants)");
    Sector s = make_sector(unrank_a, unrank_b, rank_a, rank_b, na, nb, ncas);
    hipLaunchKernelGGL(sector_pairs_kernel, dim3(n_gates + 2), dim3(1024), 0, (hipStream_t)stream, gates, n_gates, s,
                       pairs);
    OOVQE_CHECK_LAUNCH("sector_pairs");
    return 0;
}

static size_t circuit_pl_lds(int Dc, int n_gates)
{
    return ((size_t)Dc + 2 * n_gates) * sizeof(double) + (size_t)n_gates * sizeof(int) + 8;
}

static size_t adjoint_pl_lds(int Dc, int n_gates)
{
    return ((size_t)2 * Dc + 2 * n_gates + (size_t)n_gates * (SEC_PL_THREADS / 64)) * sizeof(double) +
           3 * (size_t)n_gates * sizeof(int) + 8;
}

extern "C" int oovqe_sector_state(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                  int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                  const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                  int na, int nb, int batch, double* psi_c, double* psi_dense,
                                  oovqe_stream_t stream)
{
    return oovqe_sector_state_pl(theta, n_theta, gates, n_gates, ncas, init_index, unrank_a, unrank_b, rank_a, rank_b,
                                 na, nb, batch, nullptr, 0, psi_c, psi_dense, stream);
}

// The same from the pair lists of oovqe_sector_pairs (pairs == NULL: the gate sweep above).  max_pairs = the
// largest per-gate count (the caller read the counts once).
extern "C" int oovqe_sector_state_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                     int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                     const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                     int na, int nb, int batch, const uint32_t* pairs, int max_pairs,
                                     double* psi_c, double* psi_dense, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && unrank_a && unrank_b && rank_a && rank_b && psi_c,
                  "sector_state: null pointer");
    OOVQE_REQUIRE(ncas >= 1 && ncas <= 13 && na >= 1 && nb >= 1 && batch >= 1 && n_gates >= 1,
                  "sector_state: bad sizes");
    const int Dc = na * nb;
    hipStream_t st = (hipStream_t)stream;
    Sector s = make_sector(unrank_a, unrank_b, rank_a, rank_b, na, nb, ncas);
    if (pairs) {
        const size_t lb = circuit_pl_lds(Dc, n_gates);
        OOVQE_REQUIRE(Dc <= 32767 && lb <= 160 * 1024 && max_pairs >= 0 && max_pairs <= 16 * SEC_PL_THREADS,
                      "sector_state: pair lists for %d determinants (%zu B LDS, %d pairs)", Dc, lb, max_pairs);
        int rc;
#define OOVQE_SEC_CPL(MP)                                                                          \
    do {                                                                                           \
        if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_circuit_pl_kernel<MP>, lb))) return rc; \
        hipLaunchKernelGGL(sector_circuit_pl_kernel<MP>, dim3(batch), dim3(SEC_PL_THREADS), lb, st, theta,  \
                           n_theta, gates, n_gates, s, init_index, pairs, psi_c);                  \
    } while (0)
        const int mp = (max_pairs + SEC_PL_THREADS - 1) / SEC_PL_THREADS;
        if (mp <= 1) OOVQE_SEC_CPL(1);
        else if (mp <= 2) OOVQE_SEC_CPL(2);
        else if (mp <= 4) OOVQE_SEC_CPL(4);
        else if (mp <= 8) OOVQE_SEC_CPL(8);
        else OOVQE_SEC_CPL(16);
#undef OOVQE_SEC_CPL
        OOVQE_CHECK_LAUNCH("sector_state/pairs");
    } else {
    const size_t lds_bytes = circuit_lds(na, nb, ncas, n_gates);
    OOVQE_REQUIRE(lds_bytes <= 160 * 1024 && Dc <= SEC_MAXIT * SEC_THREADS,
                  "sector_state: sector of %d determinants needs %zu B LDS", Dc, lds_bytes);
    const int nit = (Dc + SEC_THREADS - 1) / SEC_THREADS;
#define OOVQE_SEC_CIRC(MI)                                                                         \
    do {                                                                                           \
        static bool attr_done = false;                                                             \
        if (!attr_done) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)sector_circuit_kernel<MI>,             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                               160 * 1024);                                        \
            if (e != hipSuccess) {                                                                 \
                oovqe_set_error("sector_state: hipFuncSetAttribute: %s", hipGetErrorString(e));    \
                return OOVQE_ERR_HIP;                                                              \
            }                                                                                      \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(sector_circuit_kernel<MI>, dim3(batch), dim3(SEC_THREADS), lds_bytes,   \
                           st, theta, n_theta, gates, n_gates, s, init_index, psi_c);              \
    } while (0)
    if (nit <= 1) OOVQE_SEC_CIRC(1);
    else if (nit <= 2) OOVQE_SEC_CIRC(2);
    else if (nit <= 5) OOVQE_SEC_CIRC(5);
    else OOVQE_SEC_CIRC(8);
#undef OOVQE_SEC_CIRC
    OOVQE_CHECK_LAUNCH("sector_state");
    }
    if (psi_dense) {
        const uint32_t D = 1u << (2 * ncas);
        OOVQE_REQUIRE(batch <= 65535, "sector_state: batch too large for dense output");
        hipLaunchKernelGGL(sector_to_dense_kernel, dim3((D + 255) / 256 < 1024 ? (D + 255) / 256 : 1024,
                                                        batch), dim3(256), 0, st, psi_c, s, D, psi_dense);
        OOVQE_CHECK_LAUNCH("sector_state/dense");
    }
    return 0;
}

// States of the circuit with up to two gates differentiated (sector_circuit_deriv_kernel): deriv [n_out][2]
// gate indices (device, -1 = none), psi_out [batch][n_out][Dc].
extern "C" int oovqe_sector_state_deriv_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                           int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                           const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                           int na, int nb, int batch, const uint32_t* pairs, int max_pairs,
                                           const int32_t* deriv, int n_out, double* psi_out, oovqe_stream_t stream);

extern "C" int oovqe_sector_state_deriv(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                        int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                        const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                        int na, int nb, int batch, const int32_t* deriv, int n_out,
                                        double* psi_out, oovqe_stream_t stream)
{
    return oovqe_sector_state_deriv_pl(theta, n_theta, gates, n_gates, ncas, init_index, unrank_a, unrank_b, rank_a,
                                       rank_b, na, nb, batch, nullptr, 0, deriv, n_out, psi_out, stream);
}

// The same from the pair lists of oovqe_sector_pairs (pairs == NULL: the gate sweep).
extern "C" int oovqe_sector_state_deriv_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                           int n_gates, int ncas, uint32_t init_index, const uint32_t* unrank_a,
                                           const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                           int na, int nb, int batch, const uint32_t* pairs, int max_pairs,
                                           const int32_t* deriv, int n_out, double* psi_out, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && unrank_a && unrank_b && rank_a && rank_b && deriv && psi_out,
                  "sector_state_deriv: null pointer");
    OOVQE_REQUIRE(ncas >= 1 && ncas <= 13 && na >= 1 && nb >= 1 && batch >= 1 && n_gates >= 1 && n_out >= 1 &&
                  n_out <= 65535 && batch <= 65535, "sector_state_deriv: bad sizes");
    const int Dc = na * nb;
    hipStream_t st = (hipStream_t)stream;
    Sector s = make_sector(unrank_a, unrank_b, rank_a, rank_b, na, nb, ncas);
    if (pairs) {
        const size_t lb = circuit_pl_lds(Dc, n_gates);
        OOVQE_REQUIRE(Dc <= 32767 && lb <= 160 * 1024 && max_pairs >= 0 && max_pairs <= 16 * SEC_PL_THREADS,
                      "sector_state_deriv: pair lists for %d determinants (%zu B LDS, %d pairs)", Dc, lb, max_pairs);
        int rc;
#define OOVQE_SEC_DPL(MP)                                                                          \
    do {                                                                                           \
        if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_circuit_deriv_pl_kernel<MP>, lb))) return rc; \
        hipLaunchKernelGGL(sector_circuit_deriv_pl_kernel<MP>, dim3(batch, n_out), dim3(SEC_PL_THREADS), lb, st, \
                           theta, n_theta, gates, n_gates, s, init_index, pairs, deriv, n_out, psi_out); \
    } while (0)
        const int mp = (max_pairs + SEC_PL_THREADS - 1) / SEC_PL_THREADS;
        if (mp <= 1) OOVQE_SEC_DPL(1);
        else if (mp <= 2) OOVQE_SEC_DPL(2);
        else if (mp <= 4) OOVQE_SEC_DPL(4);
        else if (mp <= 8) OOVQE_SEC_DPL(8);
        else OOVQE_SEC_DPL(16);
#undef OOVQE_SEC_DPL
        OOVQE_CHECK_LAUNCH("sector_state_deriv/pairs");
        return 0;
    }
    const size_t lds_bytes = circuit_lds(na, nb, ncas, n_gates);
    OOVQE_REQUIRE(lds_bytes <= 160 * 1024 && Dc <= SEC_MAXIT * SEC_THREADS,
                  "sector_state_deriv: sector of %d determinants needs %zu B LDS", Dc, lds_bytes);
    const int nit = (Dc + SEC_THREADS - 1) / SEC_THREADS;
#define OOVQE_SEC_CIRCD(MI)                                                                        \
    do {                                                                                           \
        int rc_lds = oovqe_ensure_dynamic_lds((const void*)sector_circuit_deriv_kernel<MI>, 160 * 1024); \
        if (rc_lds) return rc_lds;                                                                 \
        hipLaunchKernelGGL(sector_circuit_deriv_kernel<MI>, dim3(batch, n_out), dim3(SEC_THREADS), lds_bytes, \
                           st, theta, n_theta, gates, n_gates, s, init_index, deriv, n_out, psi_out); \
    } while (0)
    if (nit <= 1) OOVQE_SEC_CIRCD(1);
    else if (nit <= 2) OOVQE_SEC_CIRCD(2);
    else if (nit <= 5) OOVQE_SEC_CIRCD(5);
    else OOVQE_SEC_CIRCD(8);
#undef OOVQE_SEC_CIRCD
    OOVQE_CHECK_LAUNCH("sector_state_deriv");
    return 0;
}

extern "C" int64_t oovqe_sector_work_size(int ncas, int na, int nb, int batch)
{
    const int64_t Dc = (int64_t)na * nb, na2 = (int64_t)ncas * ncas;
    const int64_t MT = (na2 + 1 + 15) / 16, NT = (na2 + 15) / 16;
    // V [batch][a^2][Dc] | W12 [batch][2 a^2 + 1][Dc] | lam [batch][Dc] | R [8 splits][batch][MT*16][NT*16]
    // | M12 [a^2][2 a^2 + 1]
    return (int64_t)batch * (3 * na2 * Dc + 2 * Dc + 8 * MT * 16 * NT * 16) + 2 * na2 * na2 + na2;
}

extern "C" int oovqe_sector_rdms_tb(const double* psi_c, int ncas, const uint32_t* unrank_a,
                                    const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                    int na, int nb, int batch, const uint16_t* tabs, double* gamma, double* Gamma,
                                    double* work, oovqe_stream_t stream);

extern "C" int oovqe_sector_rdms(const double* psi_c, int ncas, const uint32_t* unrank_a,
                                 const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                 int na, int nb, int batch, double* gamma, double* Gamma, double* work,
                                 oovqe_stream_t stream)
{
    return oovqe_sector_rdms_tb(psi_c, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb, batch, nullptr, gamma, Gamma,
                                work, stream);
}

// The same with the sector's excitation tables from memory (the block oovqe_sector_pairs leaves behind its lists;
// tabs == NULL: every workgroup builds them from the strings, as oovqe_sector_rdms does).
extern "C" int oovqe_sector_rdms_tb(const double* psi_c, int ncas, const uint32_t* unrank_a,
                                    const uint32_t* unrank_b, const int32_t* rank_a, const int32_t* rank_b,
                                    int na, int nb, int batch, const uint16_t* tabs, double* gamma, double* Gamma,
                                    double* work, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(psi_c && unrank_a && unrank_b && rank_a && rank_b && gamma && Gamma && work,
                  "sector_rdms: null pointer");
    OOVQE_REQUIRE(ncas >= 1 && ncas <= 13 && batch >= 1 && batch <= 65535, "sector_rdms: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int Dc = na * nb, na2 = ncas * ncas;
    const int MT = (na2 + 1 + 15) / 16, NT = (na2 + 15) / 16;
    Sector s = make_sector(unrank_a, unrank_b, rank_a, rank_b, na, nb, ncas);
    double* V = work;                                              // [batch][a^2][Dc]
    double* R = work + (size_t)batch * (3 * (size_t)na2 * Dc + 2 * (size_t)Dc); // [splits][batch][MT*16][NT*16]
    // a^2 a multiple of 16 and the sector vector + one chunk of E_pq vectors within a workgroup's LDS: the
    // fused kernel (V never written); nsplit workgroups per state share its chunks when the batch is small
    const size_t fused_lds = sec_fused_lds_bytes(na, nb, ncas);
    const size_t rows_lds = sec_rdm_rows_lds_bytes(na, nb, ncas);
    if ((na2 == 16 || na2 == 64) && rows_lds + 8 * 256 * sizeof(double) <= 160 * 1024 && na < SEC_TAB_MAXSTR &&
        nb < SEC_TAB_MAXSTR && (size_t)(na > nb ? na : nb) * na2 * sizeof(uint16_t) <= (size_t)na2 * SEC_RCHP * 8 &&
        oovqe_opt(OOVQE_OPT_SECTOR_UNFUSED) == 0 &&
        (oovqe_opt(OOVQE_OPT_SECTOR_RDM_R3) == 2 || (oovqe_opt(OOVQE_OPT_SECTOR_RDM_R3) == 0 && batch >= 16))) {
        // round 4: row chunks in the sigma basis (sector_rdm_rows_kernel); its longer prologue (two table builds, the
        // words of the thread's beta string) costs a single state 6 us more than it saves: from 16 states on
        const int NRC = SEC_RCH / ((nb + 8) & ~7);
        const int nchunk = (na + NRC - 1) / NRC;
        int nsplit = sector_cu_count() / batch;
        if (nsplit > 8) nsplit = 8;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > nchunk) nsplit = nchunk;
        int rc;
        if (na2 == 64) {
            if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_rdm_rows_kernel<4>, rows_lds))) return rc;
            hipLaunchKernelGGL(sector_rdm_rows_kernel<4>, dim3(batch, nsplit), dim3(512), rows_lds, st, psi_c, s, batch,
                               MT, R, oovqe_opt(OOVQE_OPT_SECTOR_PROBE), tabs);
        } else {
            if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_rdm_rows_kernel<1>, rows_lds))) return rc;
            hipLaunchKernelGGL(sector_rdm_rows_kernel<1>, dim3(batch, nsplit), dim3(512), rows_lds, st, psi_c, s, batch,
                               MT, R, oovqe_opt(OOVQE_OPT_SECTOR_PROBE), tabs);
        }
        OOVQE_CHECK_LAUNCH("sector_rdms/rows");
        hipLaunchKernelGGL(sector_rdm_finish_kernel, dim3((na2 * na2 + na2 + 255) / 256, batch), dim3(256),
                           0, st, R, ncas, batch, nsplit, gamma, Gamma);
        OOVQE_CHECK_LAUNCH("sector_rdms/finish");
        return 0;
    }
    if (na2 % 16 == 0 && na2 <= 64 && fused_lds <= 140 * 1024 && na < SEC_TAB_MAXSTR && nb < SEC_TAB_MAXSTR &&
        oovqe_opt(OOVQE_OPT_SECTOR_UNFUSED) == 0) {
        const int nchunk = (Dc + SEC_CH - 1) / SEC_CH;
        int nsplit = batch >= 128 ? 1 : (batch >= 64 ? 2 : (batch >= 16 ? 4 : 8));
        if (nsplit > nchunk) nsplit = nchunk;
#define OOVQE_SEC_RDMF(NT_)                                                                        \
        do {                                                                                       \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)sector_rdm_fused_kernel<NT_>,         \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                                (int)fused_lds), "sector_rdms: hipFuncSetAttribute"); \
            hipLaunchKernelGGL(sector_rdm_fused_kernel<NT_>, dim3(batch, nsplit), dim3(512), fused_lds, st, \
                               psi_c, s, batch, MT, R, oovqe_opt(OOVQE_OPT_SECTOR_PROBE), tabs);   \
        } while (0)
        if (NT == 4) OOVQE_SEC_RDMF(4);
        else if (NT == 2) OOVQE_SEC_RDMF(2);
        else if (NT == 3) OOVQE_SEC_RDMF(3);
        else OOVQE_SEC_RDMF(1);
#undef OOVQE_SEC_RDMF
        OOVQE_CHECK_LAUNCH("sector_rdms/fused");
        hipLaunchKernelGGL(sector_rdm_finish_kernel, dim3((na2 * na2 + na2 + 255) / 256, batch), dim3(256),
                           0, st, R, ncas, batch, nsplit, gamma, Gamma);
        OOVQE_CHECK_LAUNCH("sector_rdms/finish");
        return 0;
    }
    {
        const size_t epq_lds = ((size_t)Dc + (Dc & 1)) * sizeof(double) + 2 * ((size_t)1 << ncas) * sizeof(int32_t);
        // (few states: the one-element-per-thread grid has 64 x more workgroups to fill the chip with)
        if (epq_lds <= 64 * 1024 && batch >= 8)
            hipLaunchKernelGGL(sector_epq_rows_kernel, dim3((Dc + 255) / 256, batch), dim3(256), epq_lds, st, psi_c,
                               s, V);
        else
            hipLaunchKernelGGL(sector_epq_kernel, dim3((Dc + 255) / 256, na2, batch), dim3(256), 0, st, psi_c, s, V);
    }
    OOVQE_CHECK_LAUNCH("sector_rdms/epq");
    const int nsplit = batch >= 32 ? 1 : (batch >= 8 ? 2 : 8);   // fill the chip at small batch
    // all tiles of a state in one workgroup (V read once) for the active spaces up to 8 orbitals; the
    // tile-per-workgroup kernel beyond (more accumulators than a wave's registers hold)
    const bool rows_kernel = batch * nsplit >= 32;     // few states: more, smaller workgroups fill the chip better
    // (two half-height workgroups per state, <2, 4> with grid.z = 2, measured slower: 168 VGPRs keep
    // one workgroup per CU anyway, and capped at 128 the kernel spills: 905 us against 629)
    if (rows_kernel && MT == 5 && NT == 4)           // a = 8: psi row on the vector ALUs
        hipLaunchKernelGGL((sector_gram_rows_kernel<4, 4, true>), dim3(batch, nsplit, 1), dim3(512), 0, st, psi_c, V, ncas, Dc, batch, MT, R);
    else if (rows_kernel && MT == 4 && NT == 4)
        hipLaunchKernelGGL((sector_gram_rows_kernel<4, 4, false>), dim3(batch, nsplit, 1), dim3(512), 0, st, psi_c, V, ncas, Dc, batch, MT, R);
    else if (rows_kernel && MT == 3 && NT == 3)
        hipLaunchKernelGGL((sector_gram_rows_kernel<3, 3, false>), dim3(batch, nsplit, 1), dim3(512), 0, st, psi_c, V, ncas, Dc, batch, MT, R);
    else if (rows_kernel && MT == 2 && NT == 2)
        hipLaunchKernelGGL((sector_gram_rows_kernel<2, 2, false>), dim3(batch, nsplit, 1), dim3(512), 0, st, psi_c, V, ncas, Dc, batch, MT, R);
    else if (rows_kernel && MT == 2 && NT == 1)
        hipLaunchKernelGGL((sector_gram_rows_kernel<1, 1, true>), dim3(batch, nsplit, 1), dim3(512), 0, st, psi_c, V, ncas, Dc, batch, MT, R);
    else
        hipLaunchKernelGGL(sector_gram_kernel, dim3(MT * NT, batch, nsplit), dim3(256), 0, st, psi_c, V, ncas,
                           Dc, batch, R);
    OOVQE_CHECK_LAUNCH("sector_rdms/gram");
    hipLaunchKernelGGL(sector_rdm_finish_kernel, dim3((na2 * na2 + na2 + 255) / 256, batch), dim3(256),
                       0, st, R, ncas, batch, nsplit, gamma, Gamma);
    OOVQE_CHECK_LAUNCH("sector_rdms/finish");
    return 0;
}

// dtheta[b,k] = d/dtheta_k ( c1 . gamma(theta_b) + c2 . Gamma(theta_b) ).  psi_c and the V block of
// `work` must be those of oovqe_sector_state / oovqe_sector_rdms for the same theta (same work).
// c1 [a,a], c2 [a,a,a,a] shared by the batch (c_stride = 0) or per element (c_stride = a^2+a^4 ...).
extern "C" int oovqe_sector_adjoint_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                       int n_gates, int ncas, const uint32_t* unrank_a,
                                       const uint32_t* unrank_b, const int32_t* rank_a,
                                       const int32_t* rank_b, int na, int nb, int batch,
                                       const double* psi_c, const double* c1, const double* c2,
                                       const uint32_t* pairs, int max_pairs, const uint16_t* tabs, double* work,
                                       double* dtheta, oovqe_stream_t stream);

extern "C" int oovqe_sector_adjoint(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                    int n_gates, int ncas, const uint32_t* unrank_a,
                                    const uint32_t* unrank_b, const int32_t* rank_a,
                                    const int32_t* rank_b, int na, int nb, int batch,
                                    const double* psi_c, const double* c1, const double* c2,
                                    double* work, double* dtheta, oovqe_stream_t stream)
{
    return oovqe_sector_adjoint_pl(theta, n_theta, gates, n_gates, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb,
                                   batch, psi_c, c1, c2, nullptr, 0, nullptr, work, dtheta, stream);
}

static int sector_adjoint_impl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                               const uint32_t* unrank_a, const uint32_t* unrank_b, const int32_t* rank_a,
                               const int32_t* rank_b, int na, int nb, int batch, const double* psi_c,
                               const double* c1, const double* c2, const uint32_t* pairs, int max_pairs,
                               const uint16_t* tabs_g, double* work, double* dtheta, double* lam_out,
                               oovqe_stream_t stream, long c1_bs = 0, long c2_bs = 0, int group = 1);

// The same with the reverse sweep from the pair lists of oovqe_sector_pairs (pairs == NULL: the gate sweep).
extern "C" int oovqe_sector_adjoint_pl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                       int n_gates, int ncas, const uint32_t* unrank_a,
                                       const uint32_t* unrank_b, const int32_t* rank_a,
                                       const int32_t* rank_b, int na, int nb, int batch,
                                       const double* psi_c, const double* c1, const double* c2,
                                       const uint32_t* pairs, int max_pairs, const uint16_t* tabs, double* work,
                                       double* dtheta, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && dtheta, "sector_adjoint: null pointer");
    return sector_adjoint_impl(theta, n_theta, gates, n_gates, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb, batch,
                               psi_c, c1, c2, pairs, max_pairs, tabs, work, dtheta, nullptr, stream);
}

// lam [batch][Dc] = (Hop + Hop^T) v for a stack of sector vectors v, Hop = sum c1e_pq E_pq + sum c2_pqrs E_pq E_rs the
// operator whose quadratic form is c1 . gamma(v) + c2 . Gamma(v) (the first stage of oovqe_sector_adjoint on its
// own): v^T lam(w) / 1 = the bilinear form 2 B(v, w) that second derivatives are made of.  work: oovqe_sector_work_size.
extern "C" int oovqe_sector_lambda(const double* vecs, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                                   const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                                   const double* c1, const double* c2, const uint16_t* tabs, double* work,
                                   double* lam, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(lam, "sector_lambda: null pointer");
    return sector_adjoint_impl(nullptr, 0, nullptr, 0, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb, batch, vecs, c1,
                               c2, nullptr, 0, tabs, work, nullptr, lam, stream);
}

// The same two with the CAS coefficients of a STACK OF GEOMETRIES (kUpCCD CAS(8e,8o) over the points of a Berry-phase
// loop): state b takes c1 + (b / group) c_stride, c2 + (b / group) c_stride -- e.g. columns of the packed outputs of
// oovqe_cas_eval_batch.  group = 1: the reverse sweep of every geometry's state in one call; group = 1 + n_theta:
// the operator applied to psi and its first tangents of every geometry (the theta-theta blocks).
extern "C" int oovqe_sector_geometry_coefficients_ok(int ncas, int na, int nb)
{
    const int na2 = ncas * ncas;
    const int LAp = (na + 8) & ~7;
    return (na2 == 16 || na2 == 64) && na < SEC_TAB_MAXSTR && nb < SEC_TAB_MAXSTR && LAp <= SEC_LCH &&
           sec_fused_lds_bytes(na, nb, ncas) <= 140 * 1024 && sec_lambda_lds_bytes(na, nb, ncas) <= 160 * 1024 &&
           oovqe_opt(OOVQE_OPT_SECTOR_UNFUSED) == 0 && na * nb <= 32767;
}

extern "C" int oovqe_sector_adjoint_pg(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                                       int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                                       const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                                       const double* psi_c, const double* c1, const double* c2, int64_t c_stride,
                                       const uint32_t* pairs, int max_pairs, const uint16_t* tabs, double* work,
                                       double* dtheta, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && dtheta && c_stride > 0, "sector_adjoint_pg: null pointer / stride");
    return sector_adjoint_impl(theta, n_theta, gates, n_gates, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb, batch,
                               psi_c, c1, c2, pairs, max_pairs, tabs, work, dtheta, nullptr, stream, (long)c_stride,
                               (long)c_stride, 1);
}

extern "C" int oovqe_sector_lambda_pg(const double* vecs, int ncas, const uint32_t* unrank_a, const uint32_t* unrank_b,
                                      const int32_t* rank_a, const int32_t* rank_b, int na, int nb, int batch,
                                      int group, const double* c1, const double* c2, int64_t c_stride,
                                      const uint16_t* tabs, double* work, double* lam, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(lam && c_stride > 0 && group >= 1, "sector_lambda_pg: null pointer / stride / group");
    return sector_adjoint_impl(nullptr, 0, nullptr, 0, ncas, unrank_a, unrank_b, rank_a, rank_b, na, nb, batch, vecs, c1,
                               c2, nullptr, 0, tabs, work, nullptr, lam, stream, (long)c_stride, (long)c_stride, group);
}

static int sector_adjoint_impl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int ncas,
                               const uint32_t* unrank_a, const uint32_t* unrank_b, const int32_t* rank_a,
                               const int32_t* rank_b, int na, int nb, int batch, const double* psi_c,
                               const double* c1, const double* c2, const uint32_t* pairs, int max_pairs,
                               const uint16_t* tabs_g, double* work, double* dtheta, double* lam_out,
                               oovqe_stream_t stream, long c1_bs, long c2_bs, int group)
{
    // lam_out: stop after the lambda stage and leave it there (theta, gates, dtheta unused)
    OOVQE_REQUIRE(psi_c && c1 && c2 && work && unrank_a && unrank_b && rank_a && rank_b, "sector_adjoint: null pointer");
    OOVQE_REQUIRE(ncas >= 1 && ncas <= 13 && batch >= 1 && batch <= 65535, "sector_adjoint: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int Dc = na * nb, na2 = ncas * ncas;
    const int MT = (na2 + 1 + 15) / 16, NT = (na2 + 15) / 16;
    const size_t lds_bytes = lam_out ? 0 : pairs ? adjoint_pl_lds(na * nb, n_gates)
                                                 : adjoint_lds(na, nb, ncas, n_gates, n_theta);
    OOVQE_REQUIRE(lds_bytes <= 160 * 1024, "sector_adjoint: needs %zu B LDS", lds_bytes);
    OOVQE_REQUIRE(!pairs || (na * nb <= 32767 && max_pairs >= 0 && max_pairs <= 16 * SEC_PL_THREADS),
                  "sector_adjoint: pair lists for %d determinants, %d pairs", na * nb, max_pairs);
    Sector s = make_sector(unrank_a, unrank_b, rank_a, rank_b, na, nb, ncas);
    const size_t nb_ = (size_t)batch;
    double* V = work;
    double* W12 = V + nb_ * na2 * Dc;                   // [batch][2 a^2 + 1][Dc]
    double* lam = lam_out ? lam_out : W12 + nb_ * (2 * (size_t)na2 + 1) * Dc;
    double* R = W12 + nb_ * (2 * (size_t)na2 + 1) * Dc + nb_ * Dc;
    double* M12 = R + 8 * nb_ * (size_t)(MT * 16) * (NT * 16);
    // per_set: state b takes the CAS coefficients c1 + (b / group) c1_bs, c2 + (b / group) c2_bs (a stack of geometries:
    // group = 1 for the reverse sweep, 1 + n_theta for the operator applied to psi and its tangents); one set for all
    // otherwise.  Served by the string-driven form only (its coefficient matrices per set are small: a^4 + na^2 + nb^2).
    const bool per_set = c1_bs != 0 || c2_bs != 0;
    OOVQE_REQUIRE(group >= 1 && (!per_set || batch % group == 0), "sector_adjoint: batch %d, group %d", batch, group);
    const int nset = per_set ? batch / group : 1;
    if (!per_set) {
        hipLaunchKernelGGL(sector_coeff_kernel, dim3((na2 * na2 + 255) / 256), dim3(256), 0, st, c1, c2, ncas,
                           unrank_a, unrank_b, M12, 0L, 0L, 0L);
        OOVQE_CHECK_LAUNCH("sector_adjoint/coeff");
    }
    int rc;
    uint16_t* tabs = reinterpret_cast<uint16_t*>(M12 + (size_t)na2 * na2);   // [a^2][na] | [a^2][nb], 16-bit
    const size_t fused_lds = sec_fused_lds_bytes(na, nb, ncas);
    const bool fused = na2 % 16 == 0 && na2 <= 64 && fused_lds <= 140 * 1024 && na < SEC_TAB_MAXSTR &&
                       nb < SEC_TAB_MAXSTR &&
                       oovqe_opt(OOVQE_OPT_SECTOR_UNFUSED) == 0;
    // round 4: lambda from the string-driven form (G_a Psi' + Psi' G_b^T + the mixed term in LDS): no W
    const int LAp = (na + 8) & ~7;
    const int ta16 = (na + 15) / 16 * 16, tb16 = (nb + 15) / 16 * 16;
    const size_t dense_lds = (size_t)(ta16 > (na + 3) / 4 * 4 ? ta16 : (na + 3) / 4 * 4) *
                             ((tb16 > (nb + 3) / 4 * 4 ? tb16 : (nb + 3) / 4 * 4) + 4) * sizeof(double) +
                             ((size_t)na * na + (size_t)nb * nb) * sizeof(double);
    const size_t gmat_lds = (size_t)na2 * (na > nb ? na : nb) * (sizeof(double) + sizeof(uint16_t)) + 8;
    const size_t lam_lds = sec_lambda_lds_bytes(na, nb, ncas);
    const bool string_driven = fused && (na2 == 16 || na2 == 64) && LAp <= SEC_LCH && lam_lds <= 160 * 1024 &&
                               dense_lds <= 160 * 1024 && gmat_lds <= 160 * 1024 &&
                               (size_t)na * na + (size_t)nb * nb + Dc + ((size_t)(na + nb) * na2 + 3) / 4 + ((size_t)nb * na2 + 1) / 2 + 2 <=
                                   nb_ * (2 * (size_t)na2 + 1) * Dc &&
                               // (its two small extra launches cost ~55 us: CAS(8e,8o) 230 us whatever the batch up
                               // to 32 states, where W in memory takes 175 ... 230 us; 64: 264 vs 315; 256: 485 vs 603)
                               (per_set || oovqe_opt(OOVQE_OPT_SECTOR_LAMBDA_W) >= 2 ||
                                (oovqe_opt(OOVQE_OPT_SECTOR_LAMBDA_W) == 0 && batch >= 32));
    const long g_bs = per_set ? (long)na * na + (long)nb * nb : 0, ms_bs = per_set ? (long)na2 * na2 : 0;
    OOVQE_REQUIRE(!per_set || (string_driven && (size_t)nset * (size_t)(g_bs + ms_bs) + Dc +
                                                        ((size_t)(na + nb) * na2 + 3) / 4 + ((size_t)nb * na2 + 1) / 2 + 2 <=
                                                    nb_ * (2 * (size_t)na2 + 1) * Dc),
                  "sector_adjoint: per-geometry coefficients need the string-driven form (ncas = 4 or 8)");
    if (string_driven) {
        double* Ga = W12;                                   // (the W region of the workspace is free here)
        double* Gb = Ga + (size_t)na * na;
        // per set: [G_a | G_b] of every set, then the coefficient matrices of every set; sigma and the tables behind
        double* Ms_sets = W12 + (size_t)nset * g_bs;
        double* sigma = per_set ? Ms_sets + (size_t)nset * ms_bs : Gb + (size_t)nb * nb;
        uint16_t* tabs2 = reinterpret_cast<uint16_t*>(sigma + Dc);      // [a^2][na] | [a^2][nb]
        const double* Msrc = M12;
        if (per_set) {
            hipLaunchKernelGGL(sector_coeff_kernel, dim3((na2 * na2 + 255) / 256, nset), dim3(256), 0, st, c1, c2, ncas,
                               unrank_a, unrank_b, Ms_sets, c1_bs, c2_bs, ms_bs);
            OOVQE_CHECK_LAUNCH("sector_adjoint/coeff");
            Msrc = Ms_sets;
        }
        if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_gmat_kernel, gmat_lds))) return rc;
        hipLaunchKernelGGL(sector_gmat_kernel, dim3(na + nb + (Dc + 255) / 256, nset), dim3(256), gmat_lds, st, Msrc, s,
                           Ga, Gb, sigma, tabs2, tabs_g, ms_bs, g_bs);
        OOVQE_CHECK_LAUNCH("sector_adjoint/gmat");
        if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_lambda_dense_kernel, dense_lds))) return rc;
        hipLaunchKernelGGL(sector_lambda_dense_kernel, dim3(batch), dim3(512), dense_lds, st, psi_c, Ga, Gb, sigma,
                           na, nb, lam, g_bs, group);
        OOVQE_CHECK_LAUNCH("sector_adjoint/lambda_dense");
        const int nchunk = (nb + SEC_LCH / LAp - 1) / (SEC_LCH / LAp);
        int nsplit = sector_cu_count() / batch;             // one workgroup per CU (LDS), the chip filled once
        if (nsplit > 8) nsplit = 8;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > nchunk) nsplit = nchunk;
        const size_t pipe_lds = sec_lambda_pipe_lds_bytes(na, nb, ncas);
        if (!per_set && na2 == 64 && na <= 70 && pipe_lds <= 160 * 1024 && oovqe_opt(OOVQE_OPT_SECTOR_LAMBDA_W) == 4) {
            // the phases overlapped: multiplier and helper waves, one beta string per chunk -- measured and NOT the
            // default (344 us against 311 at 256 states: see the kernel's header)
            int nsp = sector_cu_count() / batch;
            if (nsp > 8) nsp = 8;
            if (nsp < 1) nsp = 1;
            if (nsp > nb) nsp = nb;
            if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_lambda_pipe_kernel, pipe_lds))) return rc;
            hipLaunchKernelGGL(sector_lambda_pipe_kernel, dim3(batch, nsp), dim3(512), pipe_lds, st, psi_c, M12, sigma,
                               tabs2, s, lam, oovqe_opt(OOVQE_OPT_SECTOR_PROBE));
        } else if (na2 == 64) {
            if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_lambda_fused_kernel<4>, lam_lds))) return rc;
            hipLaunchKernelGGL(sector_lambda_fused_kernel<4>, dim3(batch, nsplit), dim3(512), lam_lds, st, psi_c,
                               Msrc, sigma, tabs2, s, lam, oovqe_opt(OOVQE_OPT_SECTOR_PROBE), ms_bs, group);
        } else {
            if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_lambda_fused_kernel<1>, lam_lds))) return rc;
            hipLaunchKernelGGL(sector_lambda_fused_kernel<1>, dim3(batch, nsplit), dim3(512), lam_lds, st, psi_c,
                               Msrc, sigma, tabs2, s, lam, oovqe_opt(OOVQE_OPT_SECTOR_PROBE), ms_bs, group);
        }
        OOVQE_CHECK_LAUNCH("sector_adjoint/lambda_fused");
    } else {
    if (fused) {
        // W straight from psi: the E_pq vectors are formed chunk by chunk in LDS and contracted there
        const int nchunk = (Dc + SEC_CH - 1) / SEC_CH;
        int nsplit = batch >= 128 ? 1 : (batch >= 64 ? 2 : (batch >= 16 ? 4 : 8));
        if (nsplit > nchunk) nsplit = nchunk;
#define OOVQE_SEC_WF(NT_)                                                                          \
        do {                                                                                       \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)sector_w_fused_kernel<NT_>,           \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                                (int)fused_lds), "sector_adjoint: hipFuncSetAttribute"); \
            hipLaunchKernelGGL(sector_w_fused_kernel<NT_>, dim3(batch, nsplit), dim3(512), fused_lds, st, \
                               psi_c, M12, s, W12, tabs, tabs_g);                                  \
        } while (0)
        if (NT == 4) OOVQE_SEC_WF(4);
        else if (NT == 2) OOVQE_SEC_WF(2);
        else OOVQE_SEC_WF(1);
#undef OOVQE_SEC_WF
        OOVQE_CHECK_LAUNCH("sector_adjoint/w_fused");
    } else {
        // (the unfused path needs V: the caller's oovqe_sector_rdms on the same work left it there only when
        // it took the unfused path too -- form it here)
        const size_t epq_lds = ((size_t)Dc + (Dc & 1)) * sizeof(double) + 2 * ((size_t)1 << ncas) * sizeof(int32_t);
        if (epq_lds <= 64 * 1024 && batch >= 8)
            hipLaunchKernelGGL(sector_epq_rows_kernel, dim3((Dc + 255) / 256, batch), dim3(256), epq_lds, st, psi_c,
                               s, V);
        else
            hipLaunchKernelGGL(sector_epq_kernel, dim3((Dc + 255) / 256, na2, batch), dim3(256), 0, st, psi_c, s, V);
        OOVQE_CHECK_LAUNCH("sector_adjoint/epq");
    // W[b][j][c] = sum_k Ms[k][j] V[b][k][c]: W_pq = sum_rs (c2[pq,rs] + c2[sr,qp]) V_rs (+ the one-body
    // term on the rows (p,p)) -- one pass over V, a^2 vectors out
    if ((rc = oovqe_mode_contract_batched(V, M12, W12, 1, na2, na2, Dc, na2, 0, batch,
                                          (long)na2 * Dc, 0, (long)na2 * Dc, st)))
        return rc;
    }
    if (fused && (size_t)(na + nb) * na2 * sizeof(uint16_t) <= ((size_t)na2 * na2 + na2) * sizeof(double))
        hipLaunchKernelGGL(sector_lambda_tab_kernel, dim3((Dc + 63) / 64, batch), dim3(256), 0, st, W12, tabs,
                           tabs + (size_t)na * na2, na, nb, ncas, lam);
    else
        hipLaunchKernelGGL(sector_lambda_kernel, dim3((Dc + 63) / 64, batch), dim3(256),
                           256 * sizeof(double) + 2 * ((size_t)1 << ncas) * sizeof(int32_t), st, W12, s, lam);
    OOVQE_CHECK_LAUNCH("sector_adjoint/lambda");
    }
    if (lam_out) return 0;
    if (pairs) {
#define OOVQE_SEC_APL(MP)                                                                          \
    do {                                                                                           \
        if ((rc = oovqe_ensure_dynamic_lds((const void*)sector_adjoint_pl_kernel<MP>, lds_bytes))) return rc; \
        hipLaunchKernelGGL(sector_adjoint_pl_kernel<MP>, dim3(batch), dim3(SEC_PL_THREADS), lds_bytes, st,  \
                           theta, n_theta, gates, n_gates, Dc, pairs, psi_c, lam, dtheta);         \
    } while (0)
        const int mp = (max_pairs + SEC_PL_THREADS - 1) / SEC_PL_THREADS;
        if (mp <= 1) OOVQE_SEC_APL(1);
        else if (mp <= 2) OOVQE_SEC_APL(2);
        else if (mp <= 4) OOVQE_SEC_APL(4);
        else if (mp <= 8) OOVQE_SEC_APL(8);
        else OOVQE_SEC_APL(16);
#undef OOVQE_SEC_APL
        OOVQE_CHECK_LAUNCH("sector_adjoint/sweep_pairs");
        return 0;
    }
    const int nit = (Dc + SEC_THREADS - 1) / SEC_THREADS;
#define OOVQE_SEC_ADJ(MI)                                                                          \
    do {                                                                                           \
        static bool attr_done = false;                                                             \
        if (!attr_done) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)sector_adjoint_kernel<MI>,             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                               160 * 1024);                                        \
            if (e != hipSuccess) {                                                                 \
                oovqe_set_error("sector_adjoint: hipFuncSetAttribute: %s", hipGetErrorString(e));  \
                return OOVQE_ERR_HIP;                                                              \
            }                                                                                      \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(sector_adjoint_kernel<MI>, dim3(batch), dim3(SEC_THREADS), lds_bytes,   \
                           st, theta, n_theta, gates, n_gates, s, psi_c, lam, dtheta);             \
    } while (0)
    if (nit <= 1) OOVQE_SEC_ADJ(1);
    else if (nit <= 2) OOVQE_SEC_ADJ(2);
    else if (nit <= 5) OOVQE_SEC_ADJ(5);
    else OOVQE_SEC_ADJ(8);
#undef OOVQE_SEC_ADJ
    OOVQE_CHECK_LAUNCH("sector_adjoint/sweep");
    return 0;
}
