// CAS energy / orbital-gradient path without the N^4 MO tensor.
//
// The reference transforms the full two-electron tensor (8 N^5 flop, src/auto_oo/oo_energy.py:
// 207-208,410-411) and then slices it (utils/active_space.py:147-169; oo_energy.py:258-298).
// Every slice it ever reads for the energy and the orbital gradient has at most ONE general
// index: g_mo[n, x, y, z] with x, y, z in occ+act (M = n_occ + ncas orbitals, a contiguous
// range starting at 0, moldata_pyscf.py:50-54).  So the engine forms only
//     Gm[n,x,y,z] = sum_pqrs C[p,n] C[q,x] C[r,y] C[s,z] g_ao[p,q,r,s]
// in three stages; stage 1 is the only pass over N^4 data and is HBM-bound:
//   stage 1  T2[p,q,y,z] = sum_rs C[r,y] g_ao[p,q,r,s] C[s,z]        (this file, slab kernel)
//   stage 2  Gm = contract q->x, p->n; hmo = C^T h C[:, :M]           (contract.hip kernels)
//   stage 3  c0,c1,c2,E, Fock matrices, orbital gradient              (this file, fock kernel)
#include "common.h"

int oovqe_mode_contract_impl(const double* T, const double* Cm, double* out, long A, int K, int J,
                             long B, int ldc, int last, hipStream_t st);

namespace {

// ------------------------------------------------------------------------------------------
// Stage 1: one workgroup (4 waves) per (p,q) slab G = g_ao[p,q,:,:]  (N x N, contiguous).
// Each wave takes 16-row blocks of the slab: coalesced 16-byte global loads -> its private LDS
// tile (row pitch = 2 mod 4 doubles: conflict-free ds_read_b64 A-fragments), then
//   X[r,z]  = sum_s G[r,s] C[s,z]        MFMA, A = G tile (LDS), B = C[:, :M] (LDS, shared)
//   J[y,z] += sum_r C[r,y] X[r,z]        MFMA, A = C (LDS), B = X accumulator registers as-is
// (the f64 C/D layout row=(lane>>4)+4i, col=lane&15 is exactly a B operand of k-step i).
// Algorithmic HBM bytes: 8 N^4 read + 8 N^2 M^2 written.
// ------------------------------------------------------------------------------------------
template <int ZT>
__global__ __launch_bounds__(256)
void half_transform_kernel(const double* __restrict__ g, const double* __restrict__ C,
                           double* __restrict__ T2, int N, int M, int ldG, int nrb,
                           unsigned magicN, size_t total)
{
    constexpr int LDM = 16 * (ZT | 1);
    constexpr int JD = ZT * 16;
    extern __shared__ double lds[];
    const int RT16 = nrb * 16;
    double* Cl = lds;                           // [RT16][LDM]
    double* Gall = Cl + (size_t)RT16 * LDM;     // [4][16][ldG]
    double* red = Gall + (size_t)4 * 16 * ldG;  // [JD][JD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const size_t slab = blockIdx.x;
    const size_t slab_off = slab * (size_t)N * N;
    double* Gw = Gall + (size_t)wave * 16 * ldG;

    for (int idx = tid; idx < RT16 * LDM; idx += 256) {
        const int r = idx / LDM, z = idx - r * LDM;
        Cl[idx] = (r < N && z < M) ? C[(size_t)r * N + z] : 0.0;
    }
    for (int idx = tid; idx < JD * JD; idx += 256) red[idx] = 0.0;

    d4 jacc[ZT][ZT];
#pragma unroll
    for (int y = 0; y < ZT; ++y)
#pragma unroll
        for (int z = 0; z < ZT; ++z) jacc[y][z] = d4{0.0, 0.0, 0.0, 0.0};

    const int ksteps = (N + 3) >> 2;
    const int iters = (nrb + 3) >> 2;
    for (int it = 0; it < iters; ++it) {
        const int rb = it * 4 + wave;
        const bool active = rb < nrb;
        const int r0 = rb * 16;
        __syncthreads();   // previous tile fully consumed (and Cl / red initialised)
        if (active) {
            const int rows = (N - r0) < 16 ? (N - r0) : 16;
            const int nel = rows * N;
            const size_t gs = slab_off + (size_t)r0 * N;
            const int shift = (int)(gs & 1);
            const int span = 16 * N;
            for (int e0 = lane * 2 - shift; e0 < span; e0 += 128) {
                double v0 = 0.0, v1 = 0.0;
                if (e0 + 1 >= 0 && e0 < nel) {
                    const size_t gi = gs + (size_t)(long)e0;   // even -> 16-byte aligned
                    if (e0 >= 0 && e0 + 1 < nel && gi + 1 < total) {
                        const d2 v = *reinterpret_cast<const d2*>(g + gi);
                        v0 = v.x;
                        v1 = v.y;
                    } else {
                        if (e0 >= 0) v0 = g[gi];
                        if (e0 + 1 < nel) v1 = g[gi + 1];
                    }
                }
                if (e0 >= 0) {
                    const int row = (int)__umulhi((unsigned)e0, magicN);
                    Gw[row * ldG + (e0 - row * N)] = v0;
                }
                if (e0 + 1 < span) {
                    const int e1 = e0 + 1;
                    const int row = (int)__umulhi((unsigned)e1, magicN);
                    Gw[row * ldG + (e1 - row * N)] = v1;
                }
            }
            // zero the pad columns [N, ldG)
            const int npad = ldG - N;
            for (int idx = lane; idx < 16 * npad; idx += 64) {
                const int row = idx / npad;
                Gw[row * ldG + N + (idx - row * npad)] = 0.0;
            }
        }
        __syncthreads();
        if (active) {
            d4 x[ZT];
#pragma unroll
            for (int z = 0; z < ZT; ++z) x[z] = d4{0.0, 0.0, 0.0, 0.0};
            const double* ga = Gw + lr * ldG + lq;
            const double* cb = Cl + lq * LDM + lr;
            for (int ks = 0; ks < ksteps; ++ks) {
                const double av = ga[ks * 4];
#pragma unroll
                for (int z = 0; z < ZT; ++z) x[z] = mfma_f64(av, cb[ks * 4 * LDM + z * 16], x[z]);
            }
            const double* ca = Cl + (size_t)(r0 + lq) * LDM + lr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int y = 0; y < ZT; ++y) {
                    const double av = ca[i * 4 * LDM + y * 16];
#pragma unroll
                    for (int z = 0; z < ZT; ++z) jacc[y][z] = mfma_f64(av, x[z][i], jacc[y][z]);
                }
            }
        }
    }
    // deterministic cross-wave reduction (fixed order 0,1,2,3)
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int y = 0; y < ZT; ++y)
#pragma unroll
                for (int z = 0; z < ZT; ++z)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        red[(y * 16 + lq + 4 * i) * JD + z * 16 + lr] += jacc[y][z][i];
        }
    }
    __syncthreads();
    double* dst = T2 + slab * (size_t)M * M;
    for (int idx = tid; idx < M * M; idx += 256) {
        const int y = idx / M, z = idx - y * M;
        dst[idx] = red[y * JD + z];
    }
}

// ------------------------------------------------------------------------------------------
// Stage 3: Fock matrices, CAS coefficients, energy, orbital gradient.  One workgroup per RDM
// set (set 0 = the state's RDMs; sets k>=1 = derivative RDMs d/dtheta_k).
//   FI[n,x]   = hmo[n,x] + sum_i (2 Gm[n,x,i,i] - Gm[n,i,i,x])            oo_energy.py:272-284
//   FA[n,i]   = sum_vw gam[v,w] (Gm[n,i,V,W] - 1/2 Gm[n,W,V,i])            oo_energy.py:286-298
//   F[i,n]    = 2 (FI[n,i] + FA[n,i])                                      oo_energy.py:261-263
//   F[V,n]    = sum_w FI[n,W] gam[v,w] + sum_wxy Gam[v,w,x,y] Gm[n,W,X,Y]  oo_energy.py:264-269
//   G         = 2 (F - F^T)                                                oo_energy.py:300-309
//   c0 = nuc + sum_i (hmo[i,i] + FI[i,i]); c1 = FI[P,Q]; c2 = Gm[P,Q,R,S]/2  active_space.py:147-212
//   E  = c0 + c1.gam + c2.Gam                                              oo_energy.py:195-197
// ------------------------------------------------------------------------------------------
constexpr int FOCK_THREADS = 512;

__device__ double block_reduce_sum(double v, double* scratch)
{
    // fixed-shape tree: deterministic
    const int tid = threadIdx.x;
    scratch[tid] = v;
    __syncthreads();
    for (int s = FOCK_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) scratch[tid] += scratch[tid + s];
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(FOCK_THREADS)
void fock_kernel(const double* __restrict__ Gm, const double* __restrict__ hmo,
                 const double* __restrict__ gamma, const double* __restrict__ Gamma, double nuc,
                 int N, int no, int na, const int32_t* __restrict__ kap_row,
                 const int32_t* __restrict__ kap_col, int n_kappa, double* c0, double* c1,
                 double* c2, double* E, double* fock, double* gmat, double* gvec, double* dE)
{
    extern __shared__ double lds[];
    const int M = no + na;
    const int M2 = M * M, M3 = M2 * M;
    const int na2 = na * na, na4 = na2 * na2;
    double* FI = lds;                    // [N][M]
    double* F = FI + (size_t)N * M;      // [M][N]  generalized Fock rows < M
    double* scratch = F + (size_t)M * N; // [FOCK_THREADS]
    const int tid = threadIdx.x;
    const int k = blockIdx.x;            // RDM set
    const double* gam = gamma + (size_t)k * na2;
    const double* Gam = Gamma + (size_t)k * na4;

    for (int idx = tid; idx < N * M; idx += FOCK_THREADS) {
        const int n = idx / M, x = idx - n * M;
        const double* gn = Gm + (size_t)n * M3;
        double acc = hmo[idx];
        for (int i = 0; i < no; ++i)
            acc += 2.0 * gn[x * M2 + i * M + i] - gn[i * M2 + i * M + x];
        FI[idx] = acc;
    }
    __syncthreads();

    // occupied rows
    for (int idx = tid; idx < N * no; idx += FOCK_THREADS) {
        const int n = idx / no, i = idx - n * no;
        const double* gn = Gm + (size_t)n * M3;
        double fa = 0.0;
        for (int v = 0; v < na; ++v)
            for (int w = 0; w < na; ++w) {
                const int V = no + v, W = no + w;
                fa += gam[v * na + w] * (gn[i * M2 + V * M + W] - 0.5 * gn[W * M2 + V * M + i]);
            }
        F[i * N + n] = 2.0 * ((k == 0 ? FI[n * M + i] : 0.0) + fa);
    }
    // active rows
    for (int idx = tid; idx < N * na; idx += FOCK_THREADS) {
        const int n = idx / na, v = idx - n * na;
        const double* gn = Gm + (size_t)n * M3;
        double acc = 0.0;
        for (int w = 0; w < na; ++w) acc += FI[n * M + no + w] * gam[v * na + w];
        const double* Gv = Gam + (size_t)v * na2 * na;
        for (int w = 0; w < na; ++w)
            for (int x = 0; x < na; ++x)
                for (int y = 0; y < na; ++y)
                    acc += Gv[(w * na + x) * na + y] * gn[(no + w) * M2 + (no + x) * M + no + y];
        F[(no + v) * N + n] = acc;
    }
    __syncthreads();

    for (int t = tid; t < n_kappa; t += FOCK_THREADS) {
        const int r = kap_row[t], c = kap_col[t];
        const double frc = r < M ? F[r * N + c] : 0.0;
        const double fcr = c < M ? F[c * N + r] : 0.0;
        gvec[(size_t)k * n_kappa + t] = 2.0 * (frc - fcr);
    }

    if (k == 0) {
        if (fock)
            for (int idx = tid; idx < N * N; idx += FOCK_THREADS) {
                const int m = idx / N;
                fock[idx] = m < M ? F[idx] : 0.0;
            }
        if (gmat)
            for (int idx = tid; idx < N * N; idx += FOCK_THREADS) {
                const int m = idx / N, n = idx - m * N;
                const double fmn = m < M ? F[m * N + n] : 0.0;
                const double fnm = n < M ? F[n * N + m] : 0.0;
                gmat[idx] = 2.0 * (fmn - fnm);
            }
        for (int idx = tid; idx < na2; idx += FOCK_THREADS) {
            const int p = idx / na, q = idx - p * na;
            c1[idx] = FI[(no + p) * M + no + q];
        }
        for (int idx = tid; idx < na4; idx += FOCK_THREADS) {
            int t = idx;
            const int s = t % na; t /= na;
            const int r = t % na; t /= na;
            const int q = t % na; t /= na;
            const int p = t;
            c2[idx] = 0.5 * Gm[(size_t)(no + p) * M3 + (no + q) * M2 + (no + r) * M + no + s];
        }
    }

    // energy (k == 0) or dE/dtheta_k (k >= 1)
    double part = 0.0;
    for (int idx = tid; idx < na2; idx += FOCK_THREADS) {
        const int p = idx / na, q = idx - p * na;
        part += FI[(no + p) * M + no + q] * gam[idx];
    }
    for (int idx = tid; idx < na4; idx += FOCK_THREADS) {
        int t = idx;
        const int s = t % na; t /= na;
        const int r = t % na; t /= na;
        const int q = t % na; t /= na;
        const int p = t;
        part += 0.5 * Gm[(size_t)(no + p) * M3 + (no + q) * M2 + (no + r) * M + no + s] * Gam[idx];
    }
    if (k == 0)
        for (int i = tid; i < no; i += FOCK_THREADS) part += hmo[i * M + i] + FI[i * M + i];
    const double tot = block_reduce_sum(part, scratch);
    if (tid == 0) {
        if (k == 0) {
            // c0 alone = nuc + sum_i(...): recompute the core part serially (no <= tens)
            double core = nuc;
            for (int i = 0; i < no; ++i) core += hmo[i * M + i] + FI[i * M + i];
            c0[0] = core;
            E[0] = tot + nuc;
        } else if (dE) {
            dE[k - 1] = tot;
        }
    }
}

}  // namespace

extern "C" int oovqe_cas_half_transform(const double* g_ao, const double* C, int N, int M, double* T2,
                                        oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && C && T2, "cas_half_transform: null pointer");
    OOVQE_REQUIRE(N >= 1 && M >= 1 && M <= N, "cas_half_transform: bad N=%d M=%d", N, M);
    OOVQE_REQUIRE(((uintptr_t)g_ao & 15) == 0, "cas_half_transform: g_ao must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int ZT = (M + 15) / 16;
    const int nrb = (N + 15) / 16;
    const int ldG = ((N + 3) & ~3) + 2;
    const int LDM = 16 * (ZT | 1);
    const size_t lds_bytes =
        ((size_t)nrb * 16 * LDM + (size_t)4 * 16 * ldG + (size_t)ZT * 16 * ZT * 16) * sizeof(double);
    const size_t total = (size_t)N * N * N * N;
    if (ZT > 3 || lds_bytes > 160 * 1024) {
        // large N or M: two generic contraction passes (needs an N^3 M scratch we do not have
        // here) -> report; callers route big problems through oovqe_mode_contract themselves.
        oovqe_set_error("cas_half_transform: N=%d M=%d needs %zu B of LDS (max 163840)", N, M,
                        lds_bytes);
        return OOVQE_ERR_SIZE;
    }
    const unsigned magicN = (unsigned)((0x100000000ULL + (unsigned)N - 1) / (unsigned)N);
    const unsigned grid = (unsigned)N * (unsigned)N;
#define OOVQE_LAUNCH_HALF(Z)                                                                      \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)half_transform_kernel<Z>,             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                               160 * 1024);                                       \
            if (e != hipSuccess) {                                                                \
                oovqe_set_error("cas_half_transform: hipFuncSetAttribute: %s",                    \
                                hipGetErrorString(e));                                            \
                return OOVQE_ERR_HIP;                                                             \
            }                                                                                     \
            attr_done = true;                                                                     \
        }                                                                                         \
        hipLaunchKernelGGL((half_transform_kernel<Z>), dim3(grid), dim3(256), lds_bytes, st, g_ao, \
                           C, T2, N, M, ldG, nrb, magicN, total);                                 \
    } while (0)
    if (ZT == 1) OOVQE_LAUNCH_HALF(1);
    else if (ZT == 2) OOVQE_LAUNCH_HALF(2);
    else OOVQE_LAUNCH_HALF(3);
#undef OOVQE_LAUNCH_HALF
    OOVQE_CHECK_LAUNCH("cas_half_transform");
    return 0;
}

extern "C" int oovqe_cas_finish_transform(const double* T2, const double* h_ao, const double* C,
                                          int N, int M, double* Gm, double* hmo, double* work,
                                          oovqe_stream_t stream)
{
    OOVQE_REQUIRE(T2 && h_ao && C && Gm && hmo && work, "cas_finish_transform: null pointer");
    OOVQE_REQUIRE(N >= 1 && M >= 1 && M <= N, "cas_finish_transform: bad N=%d M=%d", N, M);
    hipStream_t st = (hipStream_t)stream;
    const long m2 = (long)M * M, m3 = m2 * M;
    double* T3 = work;                 // [N][M][M*M]
    double* Y = work + (size_t)N * m3; // [N][M]
    int rc;
    // T3[p,x,(yz)] = sum_q C[q,x] T2[p,q,(yz)]
    if ((rc = oovqe_mode_contract_impl(T2, C, T3, N, N, M, m2, N, 0, st))) return rc;
    // Gm[n,(xyz)] = sum_p C[p,n] T3[p,(xyz)]
    if ((rc = oovqe_mode_contract_impl(T3, C, Gm, 1, N, N, m3, N, 0, st))) return rc;
    // Y[p,x] = sum_q h[p,q] C[q,x] ; hmo[n,x] = sum_p C[p,n] Y[p,x]
    if ((rc = oovqe_mode_contract_impl(h_ao, C, Y, N, N, M, 1, N, 1, st))) return rc;
    if ((rc = oovqe_mode_contract_impl(Y, C, hmo, 1, N, N, M, N, 0, st))) return rc;
    return 0;
}

extern "C" int oovqe_cas_energy_gradient(const double* Gm, const double* hmo, const double* gamma,
                                         const double* Gamma, int nrdm, double nuc, int N, int n_occ,
                                         int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                         int n_kappa, double* c0, double* c1, double* c2, double* E,
                                         double* fock, double* gmat, double* gvec, double* dE,
                                         oovqe_stream_t stream)
{
    OOVQE_REQUIRE(Gm && hmo && gamma && Gamma && c0 && c1 && c2 && E && gvec,
                  "cas_energy_gradient: null pointer");
    OOVQE_REQUIRE(nrdm >= 1 && N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N,
                  "cas_energy_gradient: bad sizes");
    OOVQE_REQUIRE(n_kappa == 0 || (kap_row && kap_col), "cas_energy_gradient: null index table");
    OOVQE_REQUIRE(nrdm == 1 || dE, "cas_energy_gradient: dE required when nrdm > 1");
    const int M = n_occ + ncas;
    const size_t lds_bytes = ((size_t)2 * N * M + FOCK_THREADS) * sizeof(double);
    OOVQE_REQUIRE(lds_bytes <= 160 * 1024, "cas_energy_gradient: N*M too large for LDS");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)fock_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            oovqe_set_error("cas_energy_gradient: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fock_kernel, dim3(nrdm), dim3(FOCK_THREADS), lds_bytes, (hipStream_t)stream,
                       Gm, hmo, gamma, Gamma, nuc, N, n_occ, ncas, kap_row, kap_col, n_kappa, c0, c1,
                       c2, E, fock, gmat, gvec, dE);
    OOVQE_CHECK_LAUNCH("cas_energy_gradient");
    return 0;
}
