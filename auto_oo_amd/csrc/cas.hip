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
int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);
extern "C" int oovqe_circuit_rdms_is_small(int n_qubits, int ncas, int nvec, int n_gates);

namespace {

// ------------------------------------------------------------------------------------------
// Stage 1: one WAVE per (p,q) slab G = g_ao[p,q,:,:] (N x N, contiguous, 8 N^2 bytes), 8 waves
// per workgroup sharing only the LDS copy of C[:, :M].  Per 16-column tile (s0..s0+15) of the slab:
//   Xt[s,y]  = sum_r G[r,s] C[r,y]   MFMA: A = G^T fragment (m = s, k = r), loaded from HBM
//                                     straight into the operand register: lane (l&15, l>>4) reads
//                                     G[4ks + (l>>4)][s0 + (l&15)], i.e. four fully used 128-byte
//                                     segments per wave instruction;  B = C[:, :M] from LDS.
//   Jt[z,y] += sum_s C[s,z] Xt[s,y]  MFMA: A = C from LDS; B = the Xt accumulator registers as
//                                     they stand (the f64 C/D layout row=(lane>>4)+4i, col=lane&15
//                                     IS the B operand of k-step i) - no LDS round trip.
// Jt stays in the wave's registers for the whole slab: no cross-wave reduction, no barrier after
// the prologue, every byte of g_ao requested exactly once.  The loads of chunk c+1 are issued
// before the MFMAs of chunk c (register double buffer).
// Algorithmic HBM bytes: 8 N^4 read + 8 N^2 M^2 written.
// ------------------------------------------------------------------------------------------
constexpr int HALF_WAVES = 8;

// NST > 0: the slab has exactly NST column tiles and nkc == 1; ALL its loads (NST*KCH per lane) are
// issued at kernel entry, before the prologue barrier, so the HBM latency is paid once per wave.
// NST == 0: streaming variant for large N (register double buffer, one chunk ahead).
template <int ZT, int KCH, int NST>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_transform_kernel(const double* __restrict__ g, const double* __restrict__ C,
                           double* __restrict__ T2, int N, int M, int nst, int nkc, long nslabs)
{
    constexpr int LDM = 16 * (ZT | 1);
    extern __shared__ double lds[];
    const int RT16 = nst * 16;
    double* Cl = lds;   // [RT16][LDM], zero padded

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    // blockIdx.y = geometry of a batch (stacked g_ao [G][N^4], C [G][N^2], T2 [G][N^2 M^2])
    g += (size_t)blockIdx.y * nslabs * N * N;
    C += (size_t)blockIdx.y * N * N;
    T2 += (size_t)blockIdx.y * nslabs * M * M;

    const long slab = (long)blockIdx.x * HALF_WAVES + wave;
    const bool have = slab < nslabs;
    const double* gs = g + (size_t)(have ? slab : 0) * N * N;

    d4 jt[ZT][ZT];   // [z tile][y tile]
#pragma unroll
    for (int z = 0; z < ZT; ++z)
#pragma unroll
        for (int y = 0; y < ZT; ++y) jt[z][y] = d4{0.0, 0.0, 0.0, 0.0};
    d4 xt[ZT];       // [y tile]
#pragma unroll
    for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};

    // one column tile st (s = st*16 + lr), k-steps kc*KCH .. +KCH-1 over r.  Loads are
    // unconditional on clamped addresses, masked by a select: no branches between them.
    auto load_chunk = [&](int st, int kc, double* dst) {
        const int col = st * 16 + lr;
        const int colc = col < N ? col : N - 1;
        const int r0 = kc * KCH * 4 + lq;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int r = r0 + 4 * i;
            const int rc = r < N ? r : N - 1;
            const double v = gs[(size_t)rc * N + colc];
            // multiplicative mask (the clamped load is always finite): a select here is turned
            // back into a branch around the load by the compiler, serialising the loads
            dst[i] = v * ((have && col < N && r < N) ? 1.0 : 0.0);
        }
    };
    // Rows of Cl beyond N are zero and masked A values are zero, so every k-step can be executed
    // unconditionally (straight-line MFMA stream, C fragments read ahead).
    auto compute_chunk = [&](int st, int kc, const double* a) {
        const double* cb = Cl + (size_t)(kc * KCH * 4 + lq) * LDM + lr;
        double cf[KCH][ZT];
#pragma unroll
        for (int i = 0; i < KCH; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) cf[i][y] = cb[i * 4 * LDM + y * 16];
#pragma unroll
        for (int i = 0; i < KCH; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) xt[y] = mfma_f64(a[i], cf[i][y], xt[y]);
        if (kc == nkc - 1) {
            const double* ca = Cl + (size_t)(st * 16 + lq) * LDM + lr;
            double af[4][ZT];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int z = 0; z < ZT; ++z) af[i][z] = ca[i * 4 * LDM + z * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int z = 0; z < ZT; ++z)
#pragma unroll
                    for (int y = 0; y < ZT; ++y) jt[z][y] = mfma_f64(af[i][z], xt[y][i], jt[z][y]);
#pragma unroll
            for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};
        }
    };
    auto stage_C = [&]() {
        for (int idx = tid; idx < RT16 * LDM; idx += HALF_WAVES * 64) {
            const int r = idx / LDM, z = idx - r * LDM;
            Cl[idx] = (r < N && z < M) ? C[(size_t)r * N + z] : 0.0;
        }
        __syncthreads();
    };

    if constexpr (NST > 0) {
        // Small N: no LDS, no barrier.  Each lane fetches its own C fragments straight from L2
        // (cfr[j][y] = C[4j + lq][16y + lr]), then the whole slab; all loads of the wave are in
        // flight together and every wave runs independently of the others.
        //
        // Column tiles come in PAIRS: a pair covers 32 consecutive columns, lane lr loads the two
        // adjacent columns (2lr, 2lr+1) of row 4i+lq with ONE 16-byte load (8-byte loads reach
        // only ~0.55x of the per-CU HBM rate); .x feeds the "even" tile, .y the "odd" tile, whose
        // MFMA row index m = lr therefore stands for column c0 + 2lr (+1).  An odd last tile is a
        // single 16-column tile on 8-byte loads.
        if (!have) return;
        constexpr int NP = NST / 2, NS1 = NST % 2;
        constexpr int NROW = NST * 16;                 // rows of C touched by stage 2
        constexpr int NCF = NROW / 4;
        static_assert(KCH <= NCF, "C fragments must cover every k-step");
        double cfr[NCF][ZT];
#pragma unroll
        for (int j = 0; j < NCF; ++j)
#pragma unroll
            for (int y = 0; y < ZT; ++y) {
                const int r = 4 * j + lq, z = 16 * y + lr;
                cfr[j][y] = C[(size_t)(r < N ? r : N - 1) * N + (z < M ? z : M - 1)];
            }
        d2u apair[NP > 0 ? NP : 1][KCH];
        double asing[KCH];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            const int col = pp * 32 + 2 * lr;
            const int colc = col + 1 < N ? col : (N >= 2 ? N - 2 : 0);
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const int r = 4 * i + lq;
                apair[pp][i] = *reinterpret_cast<const d2u*>(gs + (size_t)(r < N ? r : N - 1) * N + colc);
            }
        }
        if constexpr (NS1) {
            const int col = NP * 32 + lr;
            const int colc = col < N ? col : N - 1;
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const int r = 4 * i + lq;
                asing[i] = gs[(size_t)(r < N ? r : N - 1) * N + colc];
            }
        }
        // stage 2 needs C rows in the order of the MFMA row index of each tile:
        //   even tile of pair pp: row m = lq + 4i  <->  column pp*32 + 2(lq+4i)
        //   odd  tile of pair pp:                       column pp*32 + 2(lq+4i) + 1
        //   single tile         :                       column NP*32 + lq + 4i  (= cfr[NP*8 + i])
        double cpr[NP > 0 ? NP : 1][2][4][ZT];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int z = 0; z < ZT; ++z) {
                        const int col = pp * 32 + 2 * (lq + 4 * i) + half, zz = 16 * z + lr;
                        cpr[pp][half][i][z] =
                            C[(size_t)(col < N ? col : N - 1) * N + (zz < M ? zz : M - 1)];
                    }
        __builtin_amdgcn_sched_barrier(0);
        // masks (multiplicative: a select would be turned back into a branch around the load)
#pragma unroll
        for (int j = 0; j < NCF; ++j)
#pragma unroll
            for (int y = 0; y < ZT; ++y)
                cfr[j][y] *= ((4 * j + lq) < N && (16 * y + lr) < M) ? 1.0 : 0.0;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int z = 0; z < ZT; ++z)
                        cpr[pp][half][i][z] *= ((pp * 32 + 2 * (lq + 4 * i) + half) < N &&
                                                (16 * z + lr) < M) ? 1.0 : 0.0;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            const int c0 = pp * 32 + 2 * lr;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const double colok = (c0 + half < N) ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
                    const double rowok = (4 * i + lq) < N ? colok : 0.0;
                    const double av = (half == 0 ? apair[pp][i].x : apair[pp][i].y) * rowok;
#pragma unroll
                    for (int y = 0; y < ZT; ++y) xt[y] = mfma_f64(av, cfr[i][y], xt[y]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int z = 0; z < ZT; ++z) {
                        const double cz = cpr[pp][half][i][z];
#pragma unroll
                        for (int y = 0; y < ZT; ++y) jt[z][y] = mfma_f64(cz, xt[y][i], jt[z][y]);
                    }
#pragma unroll
                for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};
            }
        }
        if constexpr (NS1) {
            const double colok = (NP * 32 + lr) < N ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const double av = asing[i] * ((4 * i + lq) < N ? colok : 0.0);
#pragma unroll
                for (int y = 0; y < ZT; ++y) xt[y] = mfma_f64(av, cfr[i][y], xt[y]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    // column NP*32 + lq + 4i = row (NP*8 + i)*4 + lq of C: an existing fragment
                    const double cz = cfr[NP * 8 + i][z];
#pragma unroll
                    for (int y = 0; y < ZT; ++y) jt[z][y] = mfma_f64(cz, xt[y][i], jt[z][y]);
                }
        }
    } else {
        // chunk c = st * nkc + kc
        const int nchunks = nst * nkc;
        double acur[KCH], anext[KCH];
        load_chunk(0, 0, acur);
        stage_C();
        if (!have) return;
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) load_chunk((c + 1) / nkc, (c + 1) % nkc, anext);
            compute_chunk(c / nkc, c % nkc, acur);
#pragma unroll
            for (int i = 0; i < KCH; ++i) acur[i] = anext[i];
        }
    }

    // jt[z tile][y tile][i] = Jt[z = zt*16 + lq + 4i][y = yt*16 + lr]  ->  T2[slab][y][z]
    double* dst = T2 + (size_t)slab * M * M;
#pragma unroll
    for (int z = 0; z < ZT; ++z)
#pragma unroll
        for (int y = 0; y < ZT; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int zz = z * 16 + lq + 4 * i, yy = y * 16 + lr;
                if (yy < M && zz < M) dst[yy * M + zz] = jt[z][y][i];
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

// ------------------------------------------------------------------------------------------
// Fused stage 2b + 3 ("column" kernel): one workgroup per general index n.
//   in : U[n,q,y,z] = sum_p C[p,n] T2[p,q,y,z]   (from the K1 contraction kernel)
//   Gn[x,y,z] = sum_q C[q,x] U[n,q,y,z]          (= g_mo[n,x,y,z], kept in LDS, 8 M^3 bytes)
//   hn[x]     = sum_q (sum_p C[p,n] h[p,q]) C[q,x]
//   FI[n,x], and for every RDM set k the n-th COLUMN of the generalized Fock matrix
//   (rows m < M only; virtual rows are zero), the per-n pieces of c0 / c1 / c2 and of the energy.
// Everything the orbital gradient needs from index n is local to this workgroup, so the MO
// integrals never travel back to HBM unless the caller asks for them (Gm_out != NULL).
// ------------------------------------------------------------------------------------------
constexpr int COL_THREADS = 256;

__global__ __launch_bounds__(COL_THREADS)
void cas_column_kernel(const double* __restrict__ U, const double* __restrict__ h_ao,
                       const double* __restrict__ C, const double* __restrict__ gamma,
                       const double* __restrict__ Gamma, int nrdm, int N, int no, int na,
                       double* __restrict__ Fcol, double* __restrict__ Epart,
                       double* __restrict__ Cpart, double* __restrict__ c1, double* __restrict__ c2,
                       double* __restrict__ Gm_out, double* __restrict__ hmo_out, size_t out_stride,
                       int rdm_chunk)
{
    extern __shared__ double lds[];
    const int M = no + na, M2 = M * M, M3 = M2 * M;
    const int na2 = na * na, na3 = na2 * na, na4 = na2 * na2;
    double* Un = lds;                    // [N][M2]
    double* Cl = Un + (size_t)N * M2;    // [N][M]   C[:, :M]
    double* Gn = Cl + (size_t)N * M;     // [M3]
    double* Wn = Gn + M3;                // [N]      (C^T h)[n, :]
    double* hn = Wn + N;                 // [M]
    double* FIn = hn + M;                // [M]
    double* cn = FIn + M;                // [N]      C[:, n]
    double* hl = cn + N;                 // [4][N]   partial sums of C[:,n]^T h
    double* gml = hl + (size_t)4 * N;    // [rdm_chunk][na2]   (RDM sets are staged chunk by chunk)
    double* Gml = gml + (size_t)rdm_chunk * na2;   // [rdm_chunk][na4]
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    {   // blockIdx.y = geometry of a batch: every per-geometry array is stacked
        const size_t gi = blockIdx.y;
        U += gi * (size_t)N * N * M2;
        h_ao += gi * (size_t)N * N;
        C += gi * (size_t)N * N;
        gamma += gi * (size_t)nrdm * na2;
        Gamma += gi * (size_t)nrdm * na4;
        Fcol += gi * (size_t)nrdm * M * N;
        Epart += gi * (size_t)nrdm * N;
        Cpart += gi * (size_t)N;
        c1 += gi * out_stride;
        c2 += gi * out_stride;
        if (Gm_out) Gm_out += gi * (size_t)N * M3;
        if (hmo_out) hmo_out += gi * (size_t)N * M;
    }

    const double* Usrc = U + (size_t)n * N * M2;
    for (int idx = tid; idx < N * M2; idx += COL_THREADS) Un[idx] = Usrc[idx];
    for (int idx = tid; idx < N * M; idx += COL_THREADS) {
        const int q = idx / M, x = idx - q * M;
        Cl[idx] = C[(size_t)q * N + x];
    }
    for (int p = tid; p < N; p += COL_THREADS) cn[p] = C[(size_t)p * N + n];
    __syncthreads();
    // W[n,q] = sum_p C[p,n] h[p,q]: 4 thread groups split the p range (coalesced rows of h),
    // partial sums combined in fixed order
    {
        double* wpart = hl;                      // [4][N] scratch (hl is only used here)
        const int q = tid & 63, part = tid >> 6;
        for (int q0 = 0; q0 < N; q0 += 64) {
            const int qq = q0 + q;
            if (qq < N) {
                const int p0 = (N * part) / 4, p1 = (N * (part + 1)) / 4;
                double a0 = 0.0, a1 = 0.0;
                int pp = p0;
                for (; pp + 1 < p1; pp += 2) {
                    a0 += cn[pp] * h_ao[(size_t)pp * N + qq];
                    a1 += cn[pp + 1] * h_ao[(size_t)(pp + 1) * N + qq];
                }
                if (pp < p1) a0 += cn[pp] * h_ao[(size_t)pp * N + qq];
                wpart[part * N + qq] = a0 + a1;
            }
        }
    }
    // Gn[x,yz] = sum_q C[q,x] U[n,q,yz]: four outputs per thread share one q loop (independent
    // LDS reads in flight instead of one dependent chain per output)
    for (int base = 0; base < M3; base += 4 * COL_THREADS) {
        int xo[4], yo[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * COL_THREADS + tid;
            ok[u] = idx < M3;
            const int ii = ok[u] ? idx : 0;
            xo[u] = ii / M2;
            yo[u] = ii - xo[u] * M2;
        }
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < N; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += Cl[q * M + xo[u]] * Un[q * M2 + yo[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) Gn[base + u * COL_THREADS + tid] = acc[u];
    }
    __syncthreads();
    for (int q = tid; q < N; q += COL_THREADS)
        Wn[q] = hl[q] + hl[N + q] + hl[2 * N + q] + hl[3 * N + q];
    __syncthreads();
    if (tid < M) {
        double acc = 0.0;
        for (int q = 0; q < N; ++q) acc += Wn[q] * Cl[q * M + tid];
        hn[tid] = acc;
        double fi = acc;
        for (int i = 0; i < no; ++i) fi += 2.0 * Gn[tid * M2 + i * M + i] - Gn[i * M2 + i * M + tid];
        FIn[tid] = fi;
    }
    if (Gm_out)
        for (int idx = tid; idx < M3; idx += COL_THREADS) Gm_out[(size_t)n * M3 + idx] = Gn[idx];
    __syncthreads();
    if (hmo_out && tid < M) hmo_out[(size_t)n * M + tid] = hn[tid];

    // per-n pieces of the CAS coefficients (independent of the RDM sets)
    if (n < no) {
        if (tid == 0) Cpart[n] = hn[n] + FIn[n];
    } else if (n < M) {
        const int p = n - no;
        if (tid == 0) Cpart[n] = 0.0;
        for (int q = tid; q < na; q += COL_THREADS) c1[p * na + q] = FIn[no + q];
        for (int idx = tid; idx < na3; idx += COL_THREADS) {
            int t = idx;
            const int s = t % na; t /= na;
            const int r = t % na; t /= na;
            const int q = t;
            c2[(size_t)p * na3 + idx] = 0.5 * Gn[(no + q) * M2 + (no + r) * M + no + s];
        }
    } else if (tid == 0) {
        Cpart[n] = 0.0;
    }

    // RDM sets, rdm_chunk at a time through LDS
    for (int k0 = 0; k0 < nrdm; k0 += rdm_chunk) {
        const int kc = (nrdm - k0) < rdm_chunk ? (nrdm - k0) : rdm_chunk;
        __syncthreads();
        for (int idx = tid; idx < kc * na2; idx += COL_THREADS) gml[idx] = gamma[(size_t)k0 * na2 + idx];
        for (int idx = tid; idx < kc * na4; idx += COL_THREADS) Gml[idx] = Gamma[(size_t)k0 * na4 + idx];
        __syncthreads();
        // Fock columns: one thread per (set k, row m)
        for (int idx = tid; idx < kc * M; idx += COL_THREADS) {
            const int kl = idx / M, m = idx - kl * M, k = k0 + kl;
            const double* gam = gml + (size_t)kl * na2;
            double val;
            if (m < no) {
                double fa = 0.0;
                for (int v = 0; v < na; ++v)
                    for (int w = 0; w < na; ++w) {
                        const int V = no + v, W = no + w;
                        fa += gam[v * na + w] * (Gn[m * M2 + V * M + W] - 0.5 * Gn[W * M2 + V * M + m]);
                    }
                val = 2.0 * ((k == 0 ? FIn[m] : 0.0) + fa);
            } else {
                const int v = m - no;
                const double* Gv = Gml + (size_t)kl * na4 + (size_t)v * na3;
                double acc = 0.0;
                for (int w = 0; w < na; ++w) acc += FIn[no + w] * gam[v * na + w];
                for (int w = 0; w < na; ++w)
                    for (int x = 0; x < na; ++x)
                        for (int y = 0; y < na; ++y)
                            acc += Gv[(w * na + x) * na + y] * Gn[(no + w) * M2 + (no + x) * M + no + y];
                val = acc;
            }
            Fcol[((size_t)k * M + m) * N + n] = val;
        }
        // E_k contribution of row p = n - no (active n only), serial per set: deterministic
        for (int kl = tid; kl < kc; kl += COL_THREADS) {
            double acc = 0.0;
            if (n >= no && n < M) {
                const int p = n - no;
                const double* gam = gml + (size_t)kl * na2 + (size_t)p * na;
                const double* Gp = Gml + (size_t)kl * na4 + (size_t)p * na3;
                for (int q = 0; q < na; ++q) acc += FIn[no + q] * gam[q];
                for (int q = 0; q < na; ++q)
                    for (int r = 0; r < na; ++r)
                        for (int s2 = 0; s2 < na; ++s2)
                            acc += 0.5 * Gn[(no + q) * M2 + (no + r) * M + no + s2]
                                   * Gp[(q * na + r) * na + s2];
            }
            Epart[(size_t)(k0 + kl) * N + n] = acc;
        }
    }
}

// Final assembly: orbital-gradient vectors, energy, c0, dE/dtheta.  One workgroup.
__global__ __launch_bounds__(512)
void cas_final_kernel(const double* __restrict__ Fcol, const double* __restrict__ Epart,
                      const double* __restrict__ Cpart, double nuc, int nrdm, int N, int M,
                      const int32_t* __restrict__ kap_row, const int32_t* __restrict__ kap_col,
                      int n_kappa, double* __restrict__ c0, double* __restrict__ E,
                      double* __restrict__ gvec, double* __restrict__ dE, double* __restrict__ fock,
                      double* __restrict__ gmat, const double* __restrict__ nuc_arr,
                      size_t out_stride)
{
    const int tid = threadIdx.x;
    {   // blockIdx.x = geometry of a batch
        const size_t gi = blockIdx.x;
        Fcol += gi * (size_t)nrdm * M * N;
        Epart += gi * (size_t)nrdm * N;
        Cpart += gi * (size_t)N;
        c0 += gi * out_stride;
        E += gi * out_stride;
        gvec += gi * out_stride;
        if (dE) dE += gi * out_stride;
        if (fock) fock += gi * (size_t)N * N;
        if (gmat) gmat += gi * (size_t)N * N;
        if (nuc_arr) nuc = nuc_arr[gi];
    }
    for (long idx = tid; idx < (long)nrdm * n_kappa; idx += 512) {
        const int k = (int)(idx / n_kappa), t = (int)(idx - (long)k * n_kappa);
        const int r = kap_row[t], c = kap_col[t];
        const double* F = Fcol + (size_t)k * M * N;
        const double frc = r < M ? F[(size_t)r * N + c] : 0.0;
        const double fcr = c < M ? F[(size_t)c * N + r] : 0.0;
        gvec[idx] = 2.0 * (frc - fcr);
    }
    if (fock)
        for (int idx = tid; idx < N * N; idx += 512) {
            const int m = idx / N;
            fock[idx] = m < M ? Fcol[idx] : 0.0;
        }
    if (gmat)
        for (int idx = tid; idx < N * N; idx += 512) {
            const int m = idx / N, n = idx - m * N;
            const double fmn = m < M ? Fcol[(size_t)m * N + n] : 0.0;
            const double fnm = n < M ? Fcol[(size_t)n * N + m] : 0.0;
            gmat[idx] = 2.0 * (fmn - fnm);
        }
    if (tid < nrdm) {
        const int k = tid;
        double acc = 0.0;
        for (int n = 0; n < M; ++n) acc += Epart[(size_t)k * N + n];
        if (k == 0) {
            double core = nuc;
            for (int n = 0; n < M; ++n) core += Cpart[n];
            c0[0] = core;
            E[0] = core + acc;
        } else if (dE) {
            dE[k - 1] = acc;
        }
    }
}

}  // namespace

static int half_transform_batched(const double* g_ao, const double* C, int N, int M, double* T2,
                                  int batch, oovqe_stream_t stream);

extern "C" int oovqe_cas_half_transform(const double* g_ao, const double* C, int N, int M, double* T2,
                                        oovqe_stream_t stream)
{
    return half_transform_batched(g_ao, C, N, M, T2, 1, stream);
}

static int half_transform_batched(const double* g_ao, const double* C, int N, int M, double* T2,
                                  int batch, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && C && T2, "cas_half_transform: null pointer");
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "cas_half_transform: batch=%d", batch);
    OOVQE_REQUIRE(N >= 1 && M >= 1 && M <= N, "cas_half_transform: bad N=%d M=%d", N, M);
    hipStream_t st = (hipStream_t)stream;
    const int ZT = (M + 15) / 16;
    const int nrb = (N + 15) / 16;
    const int ksteps = (N + 3) / 4;
    const int LDM = 16 * (ZT | 1);
    const size_t lds_elems = (((size_t)nrb * 16 * LDM + 511) / 512) * 512;   // padded, see kernel
    const size_t lds_bytes = lds_elems * sizeof(double);
    OOVQE_REQUIRE(ZT <= 3, "cas_half_transform: n_occ+ncas = %d > 48 not supported", M);
    OOVQE_REQUIRE(lds_bytes <= 160 * 1024, "cas_half_transform: N=%d M=%d needs %zu B of LDS", N, M,
                  lds_bytes);
    const long nslabs = (long)N * N;
    const unsigned grid = (unsigned)((nslabs + HALF_WAVES - 1) / HALF_WAVES);
    // k-steps per register chunk: the whole row when it fits (<= 16 k-steps), else chunks of 16
    int kch = ksteps <= 4 ? 4 : ksteps <= 8 ? 8 : ksteps <= 11 ? 11 : ksteps <= 12 ? 12 : 16;
    const int nkc = (ksteps + kch - 1) / kch;
#define OOVQE_LAUNCH_HALF(Z, KC_, NS_)                                                            \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)half_transform_kernel<Z, KC_, NS_>,   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                               160 * 1024);                                       \
            if (e != hipSuccess) {                                                                \
                oovqe_set_error("cas_half_transform: hipFuncSetAttribute: %s",                    \
                                hipGetErrorString(e));                                            \
                return OOVQE_ERR_HIP;                                                             \
            }                                                                                     \
            attr_done = true;                                                                     \
        }                                                                                         \
        hipLaunchKernelGGL((half_transform_kernel<Z, KC_, NS_>), dim3(grid, batch),               \
                           dim3(HALF_WAVES * 64),                                                 \
                           lds_bytes, st, g_ao, C, T2, N, M, nrb, nkc, nslabs);                   \
    } while (0)
#define OOVQE_DISPATCH_KCH(Z)                                                                     \
    do {                                                                                          \
        /* whole-slab preload when the slab is one k-chunk deep and <= 48 loads per lane */      \
        if (kch == 4 && nrb == 1) OOVQE_LAUNCH_HALF(Z, 4, 1);                                     \
        else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_HALF(Z, 8, 2);                                \
        else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_HALF(Z, 11, 3);                              \
        else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_HALF(Z, 12, 3);                              \
        else if (kch == 4) OOVQE_LAUNCH_HALF(Z, 4, 0);                                            \
        else if (kch == 8) OOVQE_LAUNCH_HALF(Z, 8, 0);                                            \
        else if (kch == 11) OOVQE_LAUNCH_HALF(Z, 11, 0);                                          \
        else if (kch == 12) OOVQE_LAUNCH_HALF(Z, 12, 0);                                          \
        else OOVQE_LAUNCH_HALF(Z, 16, 0);                                                         \
    } while (0)
    oovqe_profile_mark_start(st);
    if (ZT == 1) OOVQE_DISPATCH_KCH(1);
    else if (ZT == 2) OOVQE_DISPATCH_KCH(2);
    else OOVQE_DISPATCH_KCH(3);
    oovqe_profile_mark_stop(st);
#undef OOVQE_DISPATCH_KCH
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

// Batched CAS path: `batch` geometries of identical shape, every per-geometry array stacked.
// Outputs c0/c1/c2/E/gvec/dE of geometry g live at pointer + g * out_stride (doubles).
static int cas_eval_batched(const double* g_ao, const double* h_ao, const double* C,
                            const double* gamma, const double* Gamma, int nrdm, double nuc,
                            const double* nuc_arr, int N, int n_occ, int ncas, const int32_t* kap_row,
                            const int32_t* kap_col, int n_kappa, double* work, double* c0, double* c1,
                            double* c2, double* E, double* gvec, double* dE, double* fock,
                            double* gmat, double* Gm, double* hmo, int batch, size_t out_stride,
                            oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && h_ao && C && gamma && Gamma && work && c0 && c1 && c2 && E && gvec,
                  "cas_eval: null pointer");
    OOVQE_REQUIRE(nrdm >= 1 && N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N,
                  "cas_eval: bad sizes");
    OOVQE_REQUIRE(n_kappa == 0 || (kap_row && kap_col), "cas_eval: null index table");
    OOVQE_REQUIRE(nrdm == 1 || dE, "cas_eval: dE required when nrdm > 1");
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "cas_eval: batch=%d", batch);
    hipStream_t st = (hipStream_t)stream;
    const int M = n_occ + ncas;
    const long m2 = (long)M * M, m3 = m2 * M;
    // workspace layout (each block stacked over the batch)
    const size_t nb = (size_t)batch;
    double* T2 = work;                                   // [G][N][N][M][M]
    double* U = T2 + nb * N * N * m2;                    // [G][N][N][M][M]
    double* Fcol = U + nb * N * N * m2;                  // [G][nrdm][M][N]
    double* Epart = Fcol + nb * nrdm * M * N;            // [G][nrdm][N]
    double* Cpart = Epart + nb * nrdm * N;               // [G][N]
    int rc;
    if ((rc = half_transform_batched(g_ao, C, N, M, T2, batch, stream))) return rc;
    // U[n,(q y z)] = sum_p C[p,n] T2[p,(q y z)]
    oovqe_profile_mark_start_l(st, 2);
    if ((rc = oovqe_mode_contract_batched(T2, C, U, 1, N, N, (long)N * m2, N, 0, batch,
                                          (long)N * N * m2, (long)N * N, (long)N * N * m2, st)))
        return rc;
    oovqe_profile_mark_stop(st);
    const size_t na2 = (size_t)ncas * ncas;
    const size_t base_bytes = ((size_t)N * m2 + (size_t)N * M + m3 + N + M + M + N + (size_t)4 * N) *
                              sizeof(double);
    const size_t set_bytes = (na2 + na2 * na2) * sizeof(double);
    OOVQE_REQUIRE(base_bytes + set_bytes <= 160 * 1024, "cas_eval: N=%d M=%d needs %zu B of LDS", N, M,
                  base_bytes + set_bytes);
    int rdm_chunk = (int)((160 * 1024 - base_bytes) / set_bytes);
    if (rdm_chunk > nrdm) rdm_chunk = nrdm;
    const size_t lds_bytes = base_bytes + (size_t)rdm_chunk * set_bytes;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)cas_column_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            oovqe_set_error("cas_eval: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    oovqe_profile_mark_start_l(st, 3);
    hipLaunchKernelGGL(cas_column_kernel, dim3(N, batch), dim3(COL_THREADS), lds_bytes, st, U, h_ao, C,
                       gamma, Gamma, nrdm, N, n_occ, ncas, Fcol, Epart, Cpart, c1, c2, Gm, hmo,
                       out_stride, rdm_chunk);
    oovqe_profile_mark_stop(st);
    OOVQE_CHECK_LAUNCH("cas_eval/column");
    oovqe_profile_mark_start_l(st, 4);
    hipLaunchKernelGGL(cas_final_kernel, dim3(batch), dim3(512), 0, st, Fcol, Epart, Cpart, nuc, nrdm,
                       N, M, kap_row, kap_col, n_kappa, c0, E, gvec, dE, fock, gmat, nuc_arr,
                       out_stride);
    oovqe_profile_mark_stop(st);
    OOVQE_CHECK_LAUNCH("cas_eval/final");
    return 0;
}

extern "C" int oovqe_cas_eval(const double* g_ao, const double* h_ao, const double* C,
                              const double* gamma, const double* Gamma, int nrdm, double nuc, int N,
                              int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                              int n_kappa, double* work, double* c0, double* c1, double* c2,
                              double* E, double* gvec, double* dE, double* fock, double* gmat,
                              double* Gm, double* hmo, oovqe_stream_t stream)
{
    return cas_eval_batched(g_ao, h_ao, C, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row,
                            kap_col, n_kappa, work, c0, c1, c2, E, gvec, dE, fock, gmat, Gm, hmo, 1, 0,
                            stream);
}

extern "C" int64_t oovqe_cas_eval_work_size(int N, int n_occ, int ncas, int nrdm)
{
    const int64_t M = n_occ + ncas;
    return 2 * (int64_t)N * N * M * M + (int64_t)nrdm * M * N + (int64_t)nrdm * N + N;
}

// ------------------------------------------------------------------------------------------
// One call = one OO-VQE evaluation: circuit (+tangents) -> RDM sets -> CAS path.
// ------------------------------------------------------------------------------------------
extern "C" int64_t oovqe_oo_eval_work_size(int n_theta, int n_gates, int n_qubits, int N, int n_occ,
                                           int ncas, int derivatives)
{
    const int64_t D = (int64_t)1 << n_qubits;
    const int64_t nvec = derivatives ? 1 + n_theta : 1;
    const int64_t na2 = (int64_t)ncas * ncas;
    int64_t w = nvec * na2 + nvec * na2 * na2;                 // gamma, Gamma
    w += oovqe_cas_eval_work_size(N, n_occ, ncas, (int)nvec);
    if (!oovqe_circuit_rdms_is_small(n_qubits, ncas, (int)nvec, n_gates))
        w += nvec * D + nvec * na2 * D;                        // psi | dpsi, V
    return w;   // per geometry; a batch of G geometries needs G times this
}

extern "C" int64_t oovqe_oo_eval_out_size(int n_theta, int n_kappa, int ncas, int derivatives)
{
    const int64_t nvec = derivatives ? 1 + n_theta : 1;
    const int64_t n_t = nvec > 1 ? nvec - 1 : 1;
    return 2 + n_t + nvec * n_kappa + (int64_t)ncas * ncas + (int64_t)ncas * ncas * ncas * ncas;
}

// One call = one OO-VQE evaluation for each of `batch` geometries (same circuit, same shapes):
// circuit (+tangents) -> RDM sets -> CAS path; 5 launches in total, whatever the batch size.
static int oo_eval_batched(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                           int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                           const double* C, double nuc, const double* nuc_arr, int N, int n_occ,
                           int ncas, const int32_t* kap_row, const int32_t* kap_col, int n_kappa,
                           int derivatives, int batch, double* work, double* out,
                           oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && g_ao && h_ao && C && work && out, "oo_eval: null pointer");
    OOVQE_REQUIRE(n_qubits == 2 * ncas, "oo_eval: n_qubits != 2*ncas");
    OOVQE_REQUIRE(batch >= 1, "oo_eval: batch=%d", batch);
    const int nvec = derivatives ? 1 + n_theta : 1;
    const size_t nb = (size_t)batch;
    const size_t D = (size_t)1 << n_qubits;
    const size_t na2 = (size_t)ncas * ncas, na4 = na2 * na2;
    double* gamma = work;                                  // [G][nvec][a^2]
    double* Gamma = gamma + nb * nvec * na2;               // [G][nvec][a^4]
    double* cas_work = Gamma + nb * nvec * na4;
    double* rest = cas_work + nb * oovqe_cas_eval_work_size(N, n_occ, ncas, nvec);
    double *psi = nullptr, *dpsi = nullptr, *rwork = nullptr;
    if (!oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates)) {
        psi = rest;                                        // [G][D]
        dpsi = psi + nb * D;                               // [G][n_theta][D]
        rwork = psi + nb * nvec * D;
    }
    oovqe_profile_mark_start_l((hipStream_t)stream, 1);
    int rc = oovqe_circuit_rdms(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index,
                                derivatives, batch, psi, derivatives ? dpsi : nullptr, gamma, Gamma,
                                rwork, stream);
    if (rc) return rc;
    oovqe_profile_mark_stop((hipStream_t)stream);
    // packed output per geometry: [c0 | E | dE (max(nvec-1,1)) | gvec (nvec x n_kappa) | c1 | c2]
    const size_t out_stride = (size_t)oovqe_oo_eval_out_size(n_theta, n_kappa, ncas, derivatives);
    const int n_t = nvec > 1 ? nvec - 1 : 1;
    double* c0 = out;
    double* E = out + 1;
    double* dE = out + 2;
    double* gvec = dE + n_t;
    double* c1 = gvec + (size_t)nvec * n_kappa;
    double* c2 = c1 + na2;
    return cas_eval_batched(g_ao, h_ao, C, gamma, Gamma, nvec, nuc, nuc_arr, N, n_occ, ncas, kap_row,
                            kap_col, n_kappa, cas_work, c0, c1, c2, E, gvec, dE, nullptr, nullptr,
                            nullptr, nullptr, batch, out_stride, stream);
}

extern "C" int oovqe_oo_eval(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                             int n_qubits, uint32_t init_index, const double* g_ao,
                             const double* h_ao, const double* C, double nuc, int N, int n_occ,
                             int ncas, const int32_t* kap_row, const int32_t* kap_col, int n_kappa,
                             int derivatives, double* work, double* out, oovqe_stream_t stream)
{
    return oo_eval_batched(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, nuc,
                           nullptr, N, n_occ, ncas, kap_row, kap_col, n_kappa, derivatives, 1, work,
                           out, stream);
}

extern "C" int oovqe_oo_eval_batch(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                   int n_gates, int n_qubits, uint32_t init_index, const double* g_ao,
                                   const double* h_ao, const double* C, const double* nuc, int N,
                                   int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                   int n_kappa, int derivatives, int batch, double* work, double* out,
                                   oovqe_stream_t stream)
{
    OOVQE_REQUIRE(nuc, "oo_eval_batch: null nuc");
    return oo_eval_batched(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, 0.0,
                           nuc, N, n_occ, ncas, kap_row, kap_col, n_kappa, derivatives, batch, work,
                           out, stream);
}

// ------------------------------------------------------------------------------------------
// Inactive / active Fock matrices from FULL MO integrals (API helpers OO_energy.fock_core /
// fock_active, reference oo_energy.py:272-298).  The evaluation path never needs them in full
// (only the columns < M, formed in cas_column_kernel); these kernels exist so that the two public
// methods return the same N x N matrices as the reference.
//   FI[m,n] = h[m,n] + sum_i (2 g[m,n,i,i] - g[m,i,i,n])
//   FA[m,n] = sum_vw gam[v,w] (g[m,n,V,W] - 1/2 g[m,W,V,n])
// ------------------------------------------------------------------------------------------
__global__ void fock_core_active_kernel(const double* __restrict__ h, const double* __restrict__ g,
                                        const double* __restrict__ gam, int N, int no, int na,
                                        double* __restrict__ FI, double* __restrict__ FA)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    const int m = idx / N, n = idx - m * N;
    const size_t n2 = (size_t)N * N, n3 = n2 * N;
    const double* gm = g + (size_t)m * n3;
    if (FI) {
        double acc = h[idx];
        for (int i = 0; i < no; ++i)
            acc += 2.0 * gm[(size_t)n * n2 + (size_t)i * N + i] - gm[(size_t)i * n2 + (size_t)i * N + n];
        FI[idx] = acc;
    }
    if (FA) {
        double acc = 0.0;
        for (int v = 0; v < na; ++v)
            for (int w = 0; w < na; ++w) {
                const int V = no + v, W = no + w;
                acc += gam[v * na + w] * (gm[(size_t)n * n2 + (size_t)V * N + W] -
                                          0.5 * gm[(size_t)W * n2 + (size_t)V * N + n]);
            }
        FA[idx] = acc;
    }
}

extern "C" int oovqe_fock_core_active(const double* h_mo, const double* g_mo, const double* gamma, int N,
                                      int n_occ, int ncas, double* fock_core, double* fock_active,
                                      oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_mo && (fock_core || fock_active), "fock_core_active: null pointer");
    OOVQE_REQUIRE(!fock_core || h_mo, "fock_core_active: h_mo required for fock_core");
    OOVQE_REQUIRE(!fock_active || gamma, "fock_core_active: gamma required for fock_active");
    OOVQE_REQUIRE(N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N, "fock_core_active: sizes");
    hipLaunchKernelGGL(fock_core_active_kernel, dim3((N * N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, h_mo, g_mo, gamma, N, n_occ, ncas, fock_core, fock_active);
    OOVQE_CHECK_LAUNCH("fock_core_active");
    return 0;
}
