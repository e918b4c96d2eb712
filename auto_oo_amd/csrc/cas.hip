// CAS energy / orbital-gradient path without the N^4 MO tensor.
//
// The reference transforms the full two-electron tensor (8 N^5 flop, src/auto_oo/oo_energy.py:
// 207-208,410-411) and then slices it (utils/active_space.py:147-169; oo_energy.py:258-298).
// Every slice it ever reads for the energy and the orbital gradient has at most ONE general
// index: g_mo[n, x, y, z] with x, y, z in occ+act (M = n_occ + ncas orbitals, a contiguous
// range starting at 0, moldata_pyscf.py:50-54).  So the engine forms only
//     Gm[n,x,y,z] = sum_pqrs C[p,n] C[q,x] C[r,y] C[s,z] g_ao[p,q,r,s]
// in three stages; stage 1 is the only pass over N^4 data and is HBM-bound:
//   stage 1  T2[p,q,y,z] = sum_rs C[r,y] g_ao[p,q,r,s] C[s,z]        (this file, slab kernels)
//   stage 2  Gm = contract q->x, p->n; hmo = C^T h C[:, :M]
//   stage 3  c0,c1,c2,E, Fock matrices, orbital gradient
// Two realisations, chosen per call by oovqe_cas_eval (cas_eval_batched):
//   T3 path (bandwidth-bound sweeps, M <= 16, N <= 48): half_transform_fused_kernel does stage 1 and
//     the q->x half of stage 2 in one persistent kernel (T2 never leaves the chip), K1 contracts
//     p->n, cas_panel_kernel does stage 3 on panels of general indices, cas_final_kernel assembles;
//   T2 path (few geometries, or larger M / N): half_transform_kernel writes T2, K1 contracts p->n,
//     cas_column_kernel contracts q->x and does stage 3 per general index, cas_final_kernel.
// The staged entry points (oovqe_cas_half_transform / _finish_transform / _energy_gradient) keep
// the three stages separate for callers that want the intermediates.
#include "common.h"
#include "circuit_small.h"
#include <type_traits>
#include <stdlib.h>

int oovqe_mode_contract_impl(const double* T, const double* Cm, double* out, long A, int K, int J,
                             long B, int ldc, int last, hipStream_t st);
int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);
int oovqe_mode_contract_batched_circ(const double* T, const double* Cm, double* out, long A, int K, int J,
                                     long B, int ldc, int last, int batch, long t_bs, long c_bs,
                                     long o_bs, hipStream_t st, const oovqe_circuit_job_t* cj);
int oovqe_contract_hosts_circuit(long A, int K, int J, long B, int last, int batch);
int oovqe_circuit_rdms_w(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int n_qubits,
                         int ncas, uint32_t init_index, int want_tangents, int batch, double* psi, double* dpsi,
                         double* gamma, double* Gamma, double* work, const double* h_ao, const double* C, int N,
                         double* Wpre, oovqe_stream_t stream);
extern "C" int oovqe_circuit_rdms(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                                  int n_qubits, int ncas, uint32_t init_index, int want_tangents, int batch,
                                  double* psi, double* dpsi, double* gamma, double* Gamma, double* work,
                                  oovqe_stream_t stream);
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
// Cache policy of the g_ao stream (gfx950 buffer-load aux bits: 1 = sc0, 2 = nt, 16 = sc1).  Every
// byte is used once, but the non-temporal hint only pays where the stream is irregular or short:
// measured (tools/half_standalone.hip, stream_standalone.hip) packed triangle 109 -> 101 us,
// r<->s triangle on the full layout 157 -> 143 us, streaming kernel with one 16-wide tile
// (M <= 16) 323 -> 293 us (N = 64) and 367 -> 327 us (N = 128); but whole-slab streams get
// slower with it (fused kernel 319 -> 339 us, p <= q slabs 177 -> 185 us, MFMA-heavy streaming
// kernels 2810 -> 3295 us).  sc0 changes nothing, sc1 is slower.
constexpr int AUX_PLAIN = 0, AUX_NT = 2;

// Slab enumeration.  sym == 0: all N^2 slabs, t = p*N + q.  sym != 0 (the caller has verified
// g[p,q,:,:] == g[q,p,:,:] exactly): only the N(N+1)/2 slabs p <= q are read, row-major over the
// upper triangle, t = p(2N - p + 1)/2 + (q - p); the result goes to T2[p,q] AND T2[q,p] (sym == 1)
// or to the packed triangle J[t] (sym == 2).
constexpr int SYM_FULL = 0, SYM_MIRROR = 1, SYM_PACKED = 2;
constexpr int OOVQE_TRI_MODE_DEFAULT = 3;      // realisation of the packed-triangle stage 1 (half_tri_batched)
__device__ __forceinline__ void tri_decode(long t, int N, int& p, int& q)
{
    const double b = 2.0 * N + 1.0;
    int pp = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    pp = pp < 0 ? 0 : (pp > N - 1 ? N - 1 : pp);
    while (pp > 0 && (long)pp * (2 * N - pp + 1) / 2 > t) --pp;
    while (pp < N - 1 && (long)(pp + 1) * (2 * N - pp) / 2 <= t) ++pp;
    p = pp;
    q = pp + (int)(t - (long)pp * (2 * N - pp + 1) / 2);
}

// NST > 0: the slab has exactly NST column tiles and nkc == 1; ALL its loads (NST*KCH per lane) are
// issued at kernel entry, before the prologue barrier, so the HBM latency is paid once per wave.
// NST == 0: streaming variant for large N (register double buffer, one chunk ahead).
template <int ZT, int KCH, int NST>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_transform_kernel(const double* __restrict__ g, const double* __restrict__ C,
                           double* __restrict__ T2, int N, int M, int nst, int nkc, long nslabs, int sym,
                           double* __restrict__ Vk)
{
    // Vk (NST > 0, sym == SYM_MIRROR only; else null): the first product of every slab p <= q as well,
    // Vk[tri(p,q)][s][y] = sum_r g[p,q,r,s] C[r,y] -- the quarter-transformed integrals the K-type
    // (exchange) blocks of the orbital Hessian start from (hessian.hip), from the same read of the slab
    constexpr int LDM = 16 * (ZT | 1);
    extern __shared__ double lds[];
    const int RT16 = nst * 16;
    double* Cl = lds;   // [RT16][LDM], zero padded

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    // blockIdx.y = geometry of a batch (stacked g_ao [G][N^4], C [G][N^2], T2 [G][N^2 M^2] or the
    // packed triangle [G][N(N+1)/2][M^2]); nslabs = slabs to process per geometry
    g += (size_t)blockIdx.y * N * N * N * N;
    C += (size_t)blockIdx.y * N * N;
    T2 += (size_t)blockIdx.y * (sym == SYM_PACKED ? (size_t)nslabs : (size_t)N * N) * M * M;
    if (Vk) Vk += (size_t)blockIdx.y * (size_t)nslabs * N * M;

    // (NST > 0: the waves are independent, the host picks the workgroup size: four waves, so that three
    // workgroups = all 12 waves the 158 registers allow are resident on a CU instead of one workgroup of 8)
    const long slab = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool have = slab < nslabs;
    long src = have ? slab : 0, out1 = src, out2 = -1;   // slab read, slab(s) written
    if (sym != SYM_FULL) {
        int p, q;
        tri_decode(src, N, p, q);
        p = __builtin_amdgcn_readfirstlane(p);
        q = __builtin_amdgcn_readfirstlane(q);
        src = (long)p * N + q;
        if (sym == SYM_MIRROR) {
            out1 = src;
            if (p != q) out2 = (long)q * N + p;
        }
    }
    const double* gs = g + (size_t)src * N * N;

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
                    // odd N: the lane whose even column is the LAST one (c0 == N-1) loaded the pair
                    // (N-2, N-1), so its value sits in .y
                    const double ev = (c0 == N - 1) ? apair[pp][i].y : apair[pp][i].x;
                    const double av = (half == 0 ? ev : apair[pp][i].y) * rowok;
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
                if (Vk) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int sc = pp * 32 + 2 * (lq + 4 * i) + half;      // column of this row of the tile
#pragma unroll
                        for (int y = 0; y < ZT; ++y)
                            if (sc < N && 16 * y + lr < M) Vk[((size_t)slab * N + sc) * M + 16 * y + lr] = xt[y][i];
                    }
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
            if (Vk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sc = NP * 32 + lq + 4 * i;
#pragma unroll
                    for (int y = 0; y < ZT; ++y)
                        if (sc < N && 16 * y + lr < M) Vk[((size_t)slab * N + sc) * M + 16 * y + lr] = xt[y][i];
                }
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
    double* dst = T2 + (size_t)out1 * M * M;
    double* dst2 = T2 + (size_t)(out2 >= 0 ? out2 : out1) * M * M;
#pragma unroll
    for (int z = 0; z < ZT; ++z)
#pragma unroll
        for (int y = 0; y < ZT; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int zz = z * 16 + lq + 4 * i, yy = y * 16 + lr;
                if (yy < M && zz < M) {
                    dst[yy * M + zz] = jt[z][y][i];
                    if (out2 >= 0) dst2[yy * M + zz] = jt[z][y][i];
                }
            }
}

// ------------------------------------------------------------------------------------------
// Stage 1 for large N (N > 48), persistent: same per-wave algorithm as half_transform_kernel, but
//   * one resident set of workgroups; wave w of the grid takes slabs w, w + S, w + 2S, ... and runs
//     ONE continuous stream of (slab, column tile, k-chunk) chunks over them, DEPTH-1 chunks ahead
//     of the MFMAs in registers: C is staged once per workgroup and the load pipeline never drains
//     at a slab boundary;
//   * loads go through a per-slab buffer descriptor (32-bit lane offset + SGPR row-block offset);
//     rows r >= N, columns >= N and chunks past the wave's last slab are out of range for the
//     descriptor and dropped by the hardware -- no clamps, no masks, no branch around a load, so
//     the s_waitcnt counts in the loop stay exact;
//   * KCH is chosen per N so that nkc * KCH wastes at most a few k-steps (half_stream_plan).
// ------------------------------------------------------------------------------------------
#ifdef OOVQE_STREAM_PROBE
__device__ long long g_stream_cyc[16];   // workgroup 0: [0] core cycles, [1] 100 MHz ticks, [2] MFMAs of wave 0
#endif
template <int ZT, int KCH, int DEPTH, bool RS, int AUXS = (ZT == 1 ? AUX_NT : AUX_PLAIN), bool PFS = true>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_stream_kernel(const double* __restrict__ g, const double* __restrict__ C,
                        double* __restrict__ T2, int N, int M, int nst, int nkc, long nslabs, int sym)
{
    constexpr int LDM = 16 * (ZT | 1);
    extern __shared__ double lds[];
    const int RT16 = nst * 16;
    double* Cl = lds;   // [RT16][LDM], zero padded
    // RS (integrals symmetric under r <-> s, every slab a symmetric matrix): column tile S of a slab
    // only reads the rows of the 16-row tiles R <= S, the diagonal tile weighted 1/2.  That sum is
    // Z with J = Z + Z^T (the tile pair (R, S), R < S, contributes Y and its mirror image Y^T), so
    // the stream carries 54 % of the bytes and 65 % of the first-product MFMAs at N = 200 and the
    // slab's M x M result is symmetrised in registers (through a 16 x 17 LDS patch per wave) before
    // it is stored.  No repacking of the integrals.

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
#ifdef OOVQE_STREAM_PROBE
    const long long pc0 = __builtin_readcyclecounter(), pw0 = wall_clock64();
    long long n_mfma = 0;
#endif
    double* patch = lds + (((size_t)RT16 * LDM + 511) / 512) * 512 + (size_t)wave * (16 * 17);   // (RS)
    g += (size_t)blockIdx.y * N * N * N * N;
    C += (size_t)blockIdx.y * N * N;
    T2 += (size_t)blockIdx.y * (sym == SYM_PACKED ? (size_t)nslabs : (size_t)N * N) * M * M;

    const long stride = (long)gridDim.x * HALF_WAVES;
    const long slab0 = (long)blockIdx.x * HALF_WAVES + wave;
    const int n_mine = slab0 < nslabs ? (int)((nslabs - slab0 + stride - 1) / stride) : 0;
    const int nchunks = nst * nkc;
    const unsigned slab_bytes = (unsigned)((size_t)N * N * sizeof(double));
    const unsigned rowblk_bytes = (unsigned)(4 * N * sizeof(double));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));

    d4 jt[ZT][ZT];   // [z tile][y tile]
#pragma unroll
    for (int z = 0; z < ZT; ++z)
#pragma unroll
        for (int y = 0; y < ZT; ++y) jt[z][y] = d4{0.0, 0.0, 0.0, 0.0};
    d4 xt[ZT];       // [y tile]
#pragma unroll
    for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};

    // load side of the chunk stream
    auto tile_chunks = [&](int stile) {   // k-chunks column tile stile needs
        if (!RS) return nkc;
        const int c = ((stile + 1) * 4 + KCH - 1) / KCH;
        return c < nkc ? c : nkc;
    };
    int lk = 0, lst = 0, lkc = 0;
    auto issue = [&](double (&dst)[KCH]) {
#if defined(OOVQE_STREAM_PROBE) && OOVQE_STREAM_PROBE == 3
        // tools/stream_probe.hip: the stream without its loads (the first chunks' operands for ever)
        if (lk > 0) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) asm volatile("" : "+v"(dst[i]));
            if (++lkc == tile_chunks(lst)) {
                lkc = 0;
                if (++lst == nst) { lst = 0; ++lk; }
            }
            return;
        }
#endif
        long sl = lk < n_mine ? slab0 + (long)lk * stride : 0;
        if (sym != SYM_FULL) {
            int p, q;
            tri_decode(sl, N, p, q);
            sl = (long)__builtin_amdgcn_readfirstlane(p) * N + __builtin_amdgcn_readfirstlane(q);
        }
        unsigned rows_bytes = slab_bytes;   // RS: rows past the diagonal tile are out of range
        if (RS) {
            const unsigned lim = (unsigned)(lst + 1) * 4u * rowblk_bytes;
            rows_bytes = lim < slab_bytes ? lim : slab_bytes;
        }
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(g) + (size_t)sl * N * N, 0,
            lk < n_mine ? (int)rows_bytes : 0, 0x00020000);
        const int col = lst * 16 + lr;
        const unsigned vo = col < N ? (unsigned)((lq * N + col) * sizeof(double)) : 0x7fffffffu;
        const unsigned sb = (unsigned)lkc * KCH * rowblk_bytes;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, sb + i * rowblk_bytes, AUXS);
            dst[i] = __builtin_bit_cast(double, v);
        }
        if (++lkc == tile_chunks(lst)) {
            lkc = 0;
            if (++lst == nst) { lst = 0; ++lk; }
        }
    };
    // compute side.  Rows of Cl beyond N are zero and dropped loads return zero, so every k-step
    // runs unconditionally (straight-line MFMA stream, C fragments read ahead).
    int ck = 0, cst = 0, ckc = 0;
    // the rows of C a chunk multiplies (B operands, from LDS) are read one chunk ahead into the other half of
    // cfb (round 4: the LDS latency sat in front of every chunk's first MFMA; -9 % in half_tiles_kernel, -1...3 %
    // here: N = 200 general tensor 3 030-3 100 -> 2 990 us; nt loads, which help the 1 KB loads of the tile kernel,
    // cost this kernel's 128-byte row pieces 15 %)
    // (ZT = 3 has no registers for the second buffer: it reads them in front of the chunk, as before)
    constexpr bool PF = PFS && ZT <= 2;
    const double* cl_lane = Cl + lq * LDM + lr;
    double cfb[PF ? 2 : 1][KCH][ZT];
    auto load_cf = [&](int kc, double (&cf)[KCH][ZT]) {
        const double* cb = cl_lane + kc * (KCH * 4 * LDM);
#pragma unroll
        for (int i = 0; i < KCH; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) cf[i][y] = cb[i * 4 * LDM + y * 16];
    };
    auto compute = [&](const double (&a_in)[KCH], const double (&cf)[KCH][ZT], double (&cf_next)[KCH][ZT]) {
        double a[KCH];
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            a[i] = a_in[i];
            if (RS) a[i] *= ((ckc * KCH + i) >> 2) == cst ? 0.5 : 1.0;   // rows of the diagonal tile
        }
        if constexpr (PF) load_cf(ckc + 1 == tile_chunks(cst) ? 0 : ckc + 1, cf_next);
        else load_cf(ckc, cf_next);                                     // (cf_next and cf are the one buffer)
#pragma unroll
        for (int i = 0; i < KCH; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) xt[y] = mfma_f64(a[i], cf[i][y], xt[y]);
        if (++ckc == tile_chunks(cst)) {
            ckc = 0;
            const double* ca = Cl + (size_t)(cst * 16 + lq) * LDM + lr;
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
            if (++cst == nst) {
                cst = 0;
                if (RS) {
                    // J = Z + Z^T: tile (z, y) takes the transpose of tile (y, z)
                    auto transposed = [&](const d4& t) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) patch[(lq + 4 * i) * 17 + lr] = t[i];
                        d4 r;
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = patch[lr * 17 + lq + 4 * i];
                        return r;
                    };
#pragma unroll
                    for (int z = 0; z < ZT; ++z)
#pragma unroll
                        for (int y = z; y < ZT; ++y) {
                            const d4 tzy = transposed(jt[z][y]);
                            if (y == z) {
                                jt[z][z] += tzy;
                            } else {
                                const d4 tyz = transposed(jt[y][z]);
                                jt[z][y] += tyz;
                                jt[y][z] += tzy;
                            }
                        }
                }
                // jt[z tile][y tile][i] = Jt[z = zt*16 + lq + 4i][y = yt*16 + lr] -> T2[slab][y][z]
                if (ck < n_mine) {
                    long out1 = slab0 + (long)ck * stride, out2 = -1;
                    if (sym == SYM_MIRROR) {
                        int p, q;
                        tri_decode(out1, N, p, q);
                        p = __builtin_amdgcn_readfirstlane(p);
                        q = __builtin_amdgcn_readfirstlane(q);
                        out1 = (long)p * N + q;
                        if (p != q) out2 = (long)q * N + p;
                    }
                    double* dst = T2 + (size_t)out1 * M * M;
                    double* dst2 = T2 + (size_t)(out2 >= 0 ? out2 : out1) * M * M;
#pragma unroll
                    for (int z = 0; z < ZT; ++z)
#pragma unroll
                        for (int y = 0; y < ZT; ++y)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int zz = z * 16 + lq + 4 * i, yy = y * 16 + lr;
                                if (yy < M && zz < M) {
                                    dst[yy * M + zz] = jt[z][y][i];
                                    if (out2 >= 0) dst2[yy * M + zz] = jt[z][y][i];
                                }
                            }
                }
                ++ck;
#pragma unroll
                for (int z = 0; z < ZT; ++z)
#pragma unroll
                    for (int y = 0; y < ZT; ++y) jt[z][y] = d4{0.0, 0.0, 0.0, 0.0};
            }
        }
    };

    double ab[DEPTH][KCH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(ab[d]);
    for (int idx = tid; idx < RT16 * LDM; idx += HALF_WAVES * 64) {
        const int r = idx / LDM, z = idx - r * LDM;
        Cl[idx] = (r < N && z < M) ? C[(size_t)r * N + z] : 0.0;
    }
    __syncthreads();
    // the stream is padded to a multiple of DEPTH chunks: the padding chunks load nothing (their
    // slab index is past the wave's list) and add zeros
    int slab_chunks = nchunks;
    if (RS) {
        slab_chunks = 0;
        for (int t = 0; t < nst; ++t) slab_chunks += tile_chunks(t);
    }
    // (two turns of the ring per trip, so that the B-operand buffer alternates statically whatever DEPTH)
    const int rounds = (n_mine * slab_chunks + 2 * DEPTH - 1) / (2 * DEPTH);
    if constexpr (PF) load_cf(0, cfb[0]);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int d2 = 0; d2 < 2 * DEPTH; ++d2) {
            const int d = d2 % DEPTH;
            issue(ab[(d + DEPTH - 1) % DEPTH]);
            compute(ab[d], cfb[PF ? (d2 & 1) : 0], cfb[PF ? ((d2 + 1) & 1) : 0]);
        }
    }
#ifdef OOVQE_STREAM_PROBE
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        g_stream_cyc[0] = __builtin_readcyclecounter() - pc0;
        g_stream_cyc[1] = wall_clock64() - pw0;
        g_stream_cyc[2] = (long long)rounds * 2 * DEPTH * KCH * ZT + (long long)n_mine * nst * 4 * ZT * ZT;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Stage 1 for large N on the TILE-PACKED copy of integrals that carry both symmetry flags
// (oovqe_eri_pack, N > 48): per geometry the slabs p <= q; of each slab, column tile S = 0 .. nst-1 (16
// columns s), row tiles R = 0 .. S (16 rows r) -- the tile triangle half_stream_kernel<.., RS> reads out of
// the unpacked slabs -- each 16 x 16 tile stored as the four MFMA A-operand fragments it is consumed as
// ([k-step pair][lane][2]: one 16-byte load per lane and pair, 1 KB contiguous per wave instruction), the
// diagonal tiles already multiplied by 1/2, rows and columns beyond N zero.  Against the unpacked stream:
//   * no weights in the loop (they cost ~2 VALU instructions per k-step, and a VALU instruction does not run
//     in the shadow of a v_mfma_f64), no column / row masks: every byte of a tile is an operand;
//   * a chunk is ONE tile (4 k-steps): the work of a slab is exactly its tile triangle (91 tiles at N = 200),
//     where chunks of 10 k-steps padded every column tile to a multiple of 40 rows (+13 %);
//   * a wave-instruction fetches 1 KB of one 2 KB tile instead of four 128-byte row pieces.
// Same sums in the same order per accumulator as the RS stream except for the chunking, J = Z + Z^T formed
// the same way; the slab's result goes to T2[p,q] and T2[q,p].
// ------------------------------------------------------------------------------------------
template <int ZT, int DEPTH, int AUX = AUX_NT>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_tiles_kernel(const double* __restrict__ gt, const double* __restrict__ C, double* __restrict__ T2,
                       int N, int M, int nst, long nslabs, long geom_doubles)
{
    constexpr int LDM = 16 * (ZT | 1);
    extern __shared__ double lds[];
    const int RT16 = nst * 16;
    double* Cl = lds;   // [RT16][LDM], zero padded
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    double* patch = lds + (((size_t)RT16 * LDM + 511) / 512) * 512 + (size_t)wave * (16 * 17);
    gt += (size_t)blockIdx.y * geom_doubles;
    C += (size_t)blockIdx.y * N * N;
    T2 += (size_t)blockIdx.y * (size_t)N * N * M * M;

    const long stride = (long)gridDim.x * HALF_WAVES;
    const long slab0 = (long)blockIdx.x * HALF_WAVES + wave;
    const int n_mine = slab0 < nslabs ? (int)((nslabs - slab0 + stride - 1) / stride) : 0;
    const int slab_tiles = nst * (nst + 1) / 2;
    const unsigned slab_bytes = (unsigned)slab_tiles * 2048u;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));

    d4 jt[ZT][ZT];   // [z tile][y tile]
#pragma unroll
    for (int z = 0; z < ZT; ++z)
#pragma unroll
        for (int y = 0; y < ZT; ++y) jt[z][y] = d4{0.0, 0.0, 0.0, 0.0};
    d4 xt[ZT];       // [y tile]
#pragma unroll
    for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};

    // load side: tile lt of the wave's lk-th slab.  Tiles past the wave's last slab are out of range for
    // the descriptor and come back as zeros (no branch around a load: the s_waitcnt counts stay exact).
    int lk = 0, lt = 0;
    const unsigned vo = (unsigned)lane * 16u;
    auto issue = [&](d4& dst) {
        const long sl = lk < n_mine ? slab0 + (long)lk * stride : 0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(gt) + (size_t)sl * slab_tiles * 256, 0, lk < n_mine ? (int)slab_bytes : 0, 0x00020000);
        const unsigned sb = (unsigned)lt * 2048u;
        const v4u lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, sb, AUX);
        const v4u hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, sb + 1024u, AUX);
        struct pair_t { v4u a, b; } pr{lo, hi};
        dst = __builtin_bit_cast(d4, pr);
        if (++lt == slab_tiles) { lt = 0; ++lk; }
    };
    // compute side: tile (cR, cS) of the wave's ck-th slab.  The rows of C a tile multiplies (its B operands,
    // from LDS) are read one tile ahead, into the other half of cfb: the next tile is (cR + 1, cS), or row
    // tile 0 of the next column tile after a diagonal one.
    int ck = 0, cS = 0, cR = 0;
    double cfb[2][4][ZT];
    const double* cl_lane = Cl + lq * LDM + lr;
    auto load_cf = [&](int R, double (&cf)[4][ZT]) {
        const double* cb = cl_lane + R * (16 * LDM);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) cf[i][y] = cb[i * 4 * LDM + y * 16];
    };
    auto compute = [&](const d4& a, const double (&cf)[4][ZT], double (&cf_next)[4][ZT]) {
        load_cf(cR != cS ? cR + 1 : 0, cf_next);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < ZT; ++y) xt[y] = mfma_f64(a[i], cf[i][y], xt[y]);
        if (cR != cS) { ++cR; return; }
        // the diagonal tile closes column tile cS: second product with the rows of C it has just used
        cR = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int z = 0; z < ZT; ++z)
#pragma unroll
                for (int y = 0; y < ZT; ++y) jt[z][y] = mfma_f64(cf[i][z], xt[y][i], jt[z][y]);
#pragma unroll
        for (int y = 0; y < ZT; ++y) xt[y] = d4{0.0, 0.0, 0.0, 0.0};
        if (++cS != nst) return;
        cS = 0;
        // J = Z + Z^T: tile (z, y) takes the transpose of tile (y, z)
        auto transposed = [&](const d4& t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) patch[(lq + 4 * i) * 17 + lr] = t[i];
            d4 r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = patch[lr * 17 + lq + 4 * i];
            return r;
        };
#pragma unroll
        for (int z = 0; z < ZT; ++z)
#pragma unroll
            for (int y = z; y < ZT; ++y) {
                const d4 tzy = transposed(jt[z][y]);
                if (y == z) {
                    jt[z][z] += tzy;
                } else {
                    const d4 tyz = transposed(jt[y][z]);
                    jt[z][y] += tyz;
                    jt[y][z] += tzy;
                }
            }
        if (ck < n_mine) {
            int p, q;
            tri_decode(slab0 + (long)ck * stride, N, p, q);
            p = __builtin_amdgcn_readfirstlane(p);
            q = __builtin_amdgcn_readfirstlane(q);
            double* dst = T2 + ((size_t)p * N + q) * M * M;
            double* dst2 = T2 + ((size_t)q * N + p) * M * M;
#pragma unroll
            for (int z = 0; z < ZT; ++z)
#pragma unroll
                for (int y = 0; y < ZT; ++y)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int zz = z * 16 + lq + 4 * i, yy = y * 16 + lr;
                        if (yy < M && zz < M) {
                            dst[yy * M + zz] = jt[z][y][i];
                            if (p != q) dst2[yy * M + zz] = jt[z][y][i];
                        }
                    }
        }
        ++ck;
#pragma unroll
        for (int z = 0; z < ZT; ++z)
#pragma unroll
            for (int y = 0; y < ZT; ++y) jt[z][y] = d4{0.0, 0.0, 0.0, 0.0};
    };

    d4 ab[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(ab[d]);
    for (int idx = tid; idx < RT16 * LDM; idx += HALF_WAVES * 64) {
        const int r = idx / LDM, z = idx - r * LDM;
        Cl[idx] = (r < N && z < M) ? C[(size_t)r * N + z] : 0.0;
    }
    __syncthreads();
    static_assert(DEPTH % 2 == 0, "the B-operand double buffer alternates with the ring position");
    load_cf(0, cfb[0]);
    // the stream is padded to a multiple of DEPTH tiles: the padding tiles load nothing and add zeros
    const int rounds = (n_mine * slab_tiles + DEPTH - 1) / DEPTH;
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            issue(ab[(d + DEPTH - 1) % DEPTH]);
            compute(ab[d], cfb[d & 1], cfb[(d + 1) & 1]);
        }
    }
}

// The tile-packed copy itself: one workgroup per (slab p <= q, geometry), thread <-> output element.
__global__ __launch_bounds__(256)
void eri_tiles_pack_kernel(const double* __restrict__ g, double* __restrict__ out, int N, int nst)
{
    int p, q;
    tri_decode((long)blockIdx.x, N, p, q);
    const long slab_tiles = (long)nst * (nst + 1) / 2;
    const long tri = (long)N * (N + 1) / 2;
    const double* gs = g + (size_t)blockIdx.y * N * N * N * N + ((size_t)p * N + q) * N * N;
    double* o = out + ((size_t)blockIdx.y * tri + blockIdx.x) * slab_tiles * 256;
    for (long e = threadIdx.x; e < slab_tiles * 256; e += 256) {
        const int tile = (int)(e >> 8), rem = (int)(e & 255);
        int S = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while (S * (S + 1) / 2 > tile) --S;
        while ((S + 1) * (S + 2) / 2 <= tile) ++S;
        const int R = tile - S * (S + 1) / 2;
        const int ln = (rem >> 1) & 63, j = 2 * (rem >> 7) + (rem & 1);
        const int r = 16 * R + 4 * j + (ln >> 4), c = 16 * S + (ln & 15);
        double v = 0.0;
        if (r < N && c < N) v = gs[(size_t)r * N + c] * (R == S ? 0.5 : 1.0);
        o[e] = v;
    }
}

// ------------------------------------------------------------------------------------------
// q -> x on the packed triangle (p <-> q symmetric integrals):
//   T3[p,x,(y z)] = sum_q C[q,x] J[tri(min(p,q), max(p,q)), (y z)],   x < M
// One workgroup per (p, geometry), one thread per (y z); the coefficients are wave-uniform (scalar
// loads), every J element is read by two workgroups (rows p and q) and comes from L2 / MALL the
// second time.  N M^3 outputs, 2 N^2 M^3 flops per geometry: noise next to the N^4 pass.
// ------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(1024)
void sym_q_contract_kernel(const double* __restrict__ J, const double* __restrict__ C,
                           double* __restrict__ T3, int N, int M)
{
    // one wave per 16-wide (y z) tile: D[x, yz] = sum_q A[x, q] B[q, yz] on the matrix cores,
    // A = C^T (m = x, k = q), B = J rows (k = q, n = yz); all 2 KS loads of a lane are in flight
    // together, so the kernel costs about one memory round trip
    const int p = blockIdx.x;
    const long tri = (long)N * (N + 1) / 2;
    const int m2 = M * M;
    J += (size_t)blockIdx.y * tri * m2;
    C += (size_t)blockIdx.y * N * N;
    T3 += (size_t)blockIdx.y * N * M * m2;
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int yz = 16 * ty + lr;
    const int yzc = yz < m2 ? yz : m2 - 1;
    const int xc = lr < M ? lr : M - 1;
    double af[KS], bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int q = 4 * ks + lq;
        const int qc = q < N ? q : N - 1;
        const int lo = p < qc ? p : qc, hi = p < qc ? qc : p;
        const long t = (long)lo * (2 * N - lo + 1) / 2 + (hi - lo);
        bf[ks] = J[(size_t)t * m2 + yzc];
        af[ks] = C[(size_t)qc * N + xc];
    }
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = mfma_f64((4 * ks + lq) < N ? af[ks] : 0.0, bf[ks], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = lq + 4 * i;
        if (x < M && yz < m2) T3[((size_t)p * M + x) * m2 + yz] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------
// q -> x and p -> n in ONE launch on the packed triangle: one workgroup per (16-wide (y z) tile,
// geometry) computes
//   T3s[p][x][yz] = sum_q C[q,x] J[tri(min(p,q), max(p,q))][yz]     (MFMA, all p, kept in LDS)
//   Gm[n,x,yz]    = sum_p C[p,n] T3s[p][x][yz]                      (MFMA, B operand from LDS)
// for its slice of (y z): the slice of J is used by this workgroup only (its second use, as a
// column entry, comes from L2), T3 never goes to memory, and the small-circuit workgroups ride
// along as the extra grid column blockIdx.x == nty (as they do on K1, contract.hip).
// ------------------------------------------------------------------------------------------
// WPE = waves per SIMD the register allocation aims at: 2 (one workgroup per CU, all J rows of a
// wave loaded at once) or 4 (128 VGPRs, two workgroups per CU, two rows at a time: for grids of
// several resident rounds).
template <int KS, int WPE>
__global__ __launch_bounds__(SMALL_THREADS, WPE)
void sym_gm_kernel(const double* __restrict__ J, const double* __restrict__ C,
                   double* __restrict__ Gm, int N, int M, int packed, oovqe_circuit_job_t cj,
                   int host_circuit, int batch)
{
    // Workgroup -> (tile bx, geometry by).  1-D grid (gridDim.y == 1): workgroups are dealt to the 8
    // XCDs round-robin by their linear index, so index v = xcd + 8 (nx k + bx) puts the nx workgroups
    // of geometry by = xcd + 8 k on ONE XCD, next to each other in time: their partial writes of the
    // same Gm lines meet in that XCD's L2 instead of going to memory from three of them.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y == 1 && batch > 1) {
        const int nx = ((packed ? M * (M + 1) / 2 : M * M) + 15) / 16 + (host_circuit ? 1 : 0);
        const int v = blockIdx.x, s = v >> 3;
        bx = s % nx;
        by = (v & 7) + 8 * (s / nx);
        if (by >= batch) return;
    }
    // packed: the columns of J are the pairs y <= z (half_tri_kernel, tiled == 2); each result goes
    // to Gm[n,x,y,z] and Gm[n,x,z,y]
    extern __shared__ double lds[];
    const int m2 = M * M;
    const int ncol = packed ? M * (M + 1) / 2 : m2;
    const int nty = (ncol + 15) / 16;
    if (bx == nty) {
        if (host_circuit && by < cj.count)
            circuit_rdm_small_body(cj.theta, cj.n_theta, cj.gates, cj.n_gates, cj.n_qubits, cj.ncas,
                                   cj.init_index, cj.n_tan, nullptr, nullptr, cj.gamma, cj.Gamma,
                                   by, lds);
        return;
    }
    constexpr int NW = SMALL_THREADS / 64;
    constexpr int PW = (4 * KS + NW - 1) / NW;       // rows p per wave
    const int ty = bx;
    const long tri = (long)N * (N + 1) / 2;
    // J is tile-major [ty][t][16] (half_tri_kernel, tiled): this workgroup's slice is contiguous
    J += ((size_t)by * nty + ty) * tri * 16;
    C += (size_t)by * N * N;
    Gm += (size_t)by * N * M * m2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int LDP = M * 16 + 8;                      // row stride of T3s (doubles)
    double* T3s = lds;                               // [4 KS][LDP], rows p >= N are zero
    const int yz = 16 * ty + lr;                     // column of J
    int off1 = yz, off2 = -1;                        // positions y*M + z (and z*M + y) in Gm[n,x,:,:]
    if (packed) {
        int y, z;
        tri_decode(yz < ncol ? yz : 0, M, y, z);
        off1 = y * M + z;
        if (y != z) off2 = z * M + y;
    }

    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(C), 0, (int)((size_t)N * N * sizeof(double)), 0x00020000);
    // ---- step 1: A = C^T (m = x = lr, k = q = 4ks + lq), B = J rows (k = q, n = yz); the columns
    // beyond the last one of the last tile hold unwritten memory: they only reach output columns
    // never stored
    {
        const __amdgpu_buffer_rsrc_t jsrd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(J), 0, (int)(tri * 16 * sizeof(double)), 0x00020000);
        double af[KS];
        // triangle row index of the pair (p, q): t = base(lo) + hi with base(a) = a(2N - a + 1)/2 - a;
        // per lane and k-step: q and base(q), fixed for the whole kernel (32-bit arithmetic)
        int qk[KS], baseq[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int q = 4 * ks + lq;
            qk[ks] = q < N ? q : N - 1;
            baseq[ks] = qk[ks] * (2 * N - qk[ks] + 1) / 2 - qk[ks];
            // coefficients through the descriptor: out-of-range lanes (q >= N, x >= M) get an
            // out-of-range offset and read 0 (a select or a branch around a plain load makes the
            // compiler wait for each load separately)
            const v2u v = __builtin_amdgcn_raw_buffer_load_b64(
                csrd, (q < N && lr < M) ? (unsigned)((q * N + lr) * (int)sizeof(double)) : 0x7fffffffu, 0, 0);
            af[ks] = __builtin_bit_cast(double, v);
        }
        // PB rows of the wave at a time: PB * KS loads of a lane in flight, then their MFMA chains
        constexpr int PB = WPE >= 6 ? 1 : WPE >= 4 ? 2 : PW;
#pragma unroll
        for (int h = 0; h < PW; h += PB) {
            double bf[PB][KS];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int p = wave + NW * (h + i);
                const int pc = p < N ? p : N - 1;                       // wave-uniform
                const int basep = pc * (2 * N - pc + 1) / 2 - pc;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int t = qk[ks] < pc ? baseq[ks] + pc : basep + qk[ks];
                    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(
                        jsrd, (unsigned)((t * 16 + lr) * (int)sizeof(double)), 0, 0);
                    bf[i][ks] = __builtin_bit_cast(double, v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                if (h + i >= PW) continue;
                const int p = wave + NW * (h + i);
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = mfma_f64(af[ks], bf[i][ks], acc);
                // rows p in [N, 4 KS) are the zero padding of step 2's k range
                double* row = T3s + (size_t)(p < 4 * KS ? p : 0) * LDP + lr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = lq + 4 * j;
                    if (x < M && p < 4 * KS) row[x * 16] = p < N ? acc[j] : 0.0;
                }
            }
        }
    }
    __syncthreads();

    // ---- step 2: output tile (n tile nt, x): A = C^T (m = n = 16 nt + lr, k = p = 4ks + lq) from
    // L2, B = T3s[p][x][yz] from LDS; D rows n = 16 nt + lq + 4j, columns yz
    const int ntn = (N + 15) / 16;
    int cur_nt = -1;
    double cf[KS];
    for (int tile = wave; tile < ntn * M; tile += NW) {
        const int nt = tile / M, x = tile - nt * M;
        if (nt != cur_nt) {
            const int n = 16 * nt + lr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int pp = 4 * ks + lq;
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(
                    csrd, (pp < N && n < N) ? (unsigned)((pp * N + n) * (int)sizeof(double)) : 0x7fffffffu, 0, 0);
                cf[ks] = __builtin_bit_cast(double, v);
            }
            cur_nt = nt;
        }
        double tf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tf[ks] = T3s[(size_t)(4 * ks + lq) * LDP + x * 16 + lr];
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma_f64(cf[ks], tf[ks], acc);
        if (yz < ncol) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 16 * nt + lq + 4 * j;
                if (n < N) {
                    double* dst = Gm + ((size_t)n * M + x) * m2;
                    dst[off1] = acc[j];
                    if (off2 >= 0) dst[off2] = acc[j];
                }
            }
        }
    }
}

// J of sym_gm_kernel, tile-major [ty][t = tri(p,q)][16] over the columns (y z) (packed: the pairs y <= z), from
// half-transformed integrals T2[p,q,y,z] that are in memory already (the Hessian call: hessian.hip)
__global__ __launch_bounds__(256)
void j_from_t2_kernel(const double* __restrict__ T2, double* __restrict__ J, int N, int M, int packed)
{
    // 16 rows t per workgroup, thread <-> (row, column c of a tile); the position of every column in the
    // M x M block once per workgroup (LDS), (p, q) of a row once per thread
    __shared__ int off[256];
    const long tri = (long)N * (N + 1) / 2;
    const int m2 = M * M;
    const int ncol = packed ? M * (M + 1) / 2 : m2;
    const int nty = (ncol + 15) / 16;
    T2 += (size_t)blockIdx.y * N * N * m2;
    J += (size_t)blockIdx.y * nty * tri * 16;
    const int tid = threadIdx.x;
    if (tid < nty * 16) {
        int y = 0, z = 0, o = -1;
        if (tid < ncol) {
            if (packed) tri_decode(tid, M, y, z);
            else { y = tid / M; z = tid - y * M; }
            o = y * M + z;
        }
        off[tid] = o;
    }
    __syncthreads();
    const int c = tid & 15;
    for (long t = (long)blockIdx.x * 16 + (tid >> 4); t < tri; t += (long)gridDim.x * 16) {
        int p, q;
        tri_decode(t, N, p, q);
        const double* src = T2 + ((size_t)p * N + q) * m2;
        for (int ty = 0; ty < nty; ++ty) {
            const int o = off[ty * 16 + c];
            J[((size_t)ty * tri + t) * 16 + c] = o >= 0 ? src[o] : 0.0;
        }
    }
}

// Exact (bitwise) symmetry tests, one workgroup per (slab (p,q), geometry):
//   bit 0 of *mismatch: some g[p,q,:,:] != g[q,p,:,:];  bit 1: some g[p,q,r,s] != g[p,q,s,r]
__global__ __launch_bounds__(256)
void eri_symmetry_check_kernel(const unsigned long long* __restrict__ g, int N, int* __restrict__ mismatch)
{
    const int p = blockIdx.x, q = blockIdx.y;
    const size_t n2 = (size_t)N * N;
    const unsigned long long* a = g + (size_t)blockIdx.z * n2 * n2 + ((size_t)p * N + q) * n2;
    const unsigned long long* b = g + (size_t)blockIdx.z * n2 * n2 + ((size_t)q * N + p) * n2;
    bool bad_pq = false, bad_rs = false;
    for (size_t i = threadIdx.x; i < n2; i += 256) {
        const unsigned long long v = a[i];
        if (p < q) bad_pq |= v != b[i];
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        if (r < c) bad_rs |= v != a[(size_t)c * N + r];
    }
    if (bad_pq) atomicOr(mismatch, 1);
    if (bad_rs) atomicOr(mismatch, 2);
}

// ------------------------------------------------------------------------------------------
// Stages 1 + 2a fused, persistent and software-pipelined (M <= 16, N <= 48: a slab is one register
// chunk):   T3[c][p][x][y][z] = sum_{q in chunk c} C[q,x] * ( sum_rs C[r,y] g[p,q,r,s] C[s,z] )
//
// Why fused, and why nothing is stored inside the loop: ANY store traffic interleaved with the read
// stream costs far more than its bytes.  Measured on this kernel (64 geometries, N = 43, M = 9,
// 1.75 GB read): no stores 305-313 us; T2 written per slab (77 MB, 4 % of the bytes) 385 us; T3
// written per task (16 MB, 1 %) 345 us -- the same whether the stores are coalesced, line-aligned,
// non-temporal, aimed at an L2-resident region, issued by a dedicated wave, or kept out of the
// s_waitcnt chain; only dropping them (or thinning them) helps.  That is the signature of
// write bursts turning the HBM channels around inside the read stream, so the kernel contracts
// q -> x in the workgroup (T3 is 8 N M^3 bytes, M/N of T2), keeps its T3 rows in LDS and writes
// them in one burst after its last slab.
//
// Work split: a task is (chunk c of the q range, p); a workgroup owns the tasks bx, bx+W, ... of
// its geometry and its 8 waves take the slab positions of that task list round-robin (position j
// = task j/qc, slot j%qc), so waves stay busy across task boundaries.  Per wave: the loads of its
// next slab are issued before the MFMAs of the current one (two register buffers, loop unrolled by
// two, straight-line body); C fragments are loaded once.  Per slab the 16x16 result tile goes to
// an LDS block [slot][y*M+z] of the task (ring of nbuf blocks).  The arrival that completes a block
// (LDS counter) opens its tile jobs: one job contracts the block with C[q,x] for 16 (y,z) on the
// MFMA (A = C rows from LDS, B = block) into the LDS stage.  Every wave takes at most one open job
// per slab, so the contraction is spread over the waves (done by the completing wave alone it
// made that wave the straggler of every following task: +9 % kernel time).  A block is reused
// only after its generation counter says the previous task in it has been consumed.  One
// workgroup barrier after the prologue, one before the final T3 burst.
// LDS ordering relies on one wave's LDS instructions executing in issue order (ds_write of the
// tile, then ds_add of the arrival).  Rows/columns beyond N are dropped loads (descriptor range
// check) and zeroed C fragments, so g_ao needs no masks (it must be finite).
// nchunk > 1 (few geometries: more tasks than N per geometry) leaves partial sums per chunk; the
// p -> n contraction that follows sums them through Cdup[(c,p)][n] = C[p][n], written here.
// ------------------------------------------------------------------------------------------
template <int KCH, int NST>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_transform_fused_kernel(const double* __restrict__ g, const double* __restrict__ C,
                                 double* __restrict__ T3, double* __restrict__ Cdup, int N, int M,
                                 int nchunk, int qc, int nbuf, int ldb)
{
    constexpr int NP = NST / 2, NS1 = NST % 2;
    constexpr int NPA = NP > 0 ? NP : 1;
    constexpr int NCF = NST * 4;
    static_assert(KCH <= NCF, "C fragments must cover every k-step");
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar control flow
    const int lq = lane >> 4, lr = lane & 15;
    const int M2 = M * M;
    const long nslabs = (long)N * N;
    const size_t slab_elems = (size_t)N * N;
    const int QR = (qc + 3) & ~3;                   // rows of a block (k-steps of the q contraction)
    const int CR = (nchunk - 1) * qc + QR;          // rows of the LDS copy of C[:, :M]
    double* CxL = lds;                              // [CR][16], zero padded
    double* blk = CxL + (size_t)CR * 16;            // [nbuf][QR][ldb]
    double* dump = blk + (size_t)nbuf * QR * ldb;   // [64] sink for lanes outside the M x M tile
    int* cnt = reinterpret_cast<int*>(dump + 64);   // [nbuf] arrivals at the task in block b
    int* gen = cnt + nbuf;                          // [nbuf] tasks consumed in block b
    int* job = gen + nbuf;                          // [nbuf] next tile job of block b (>= nty: closed)
    int* done = job + nbuf;                         // [nbuf] tile jobs finished
    int* tsk = done + nbuf;                         // [nbuf] task held by block b
    double* stg = dump + 64 + 8;                    // [ntask][M][M2] T3 of this workgroup's tasks
    g += (size_t)blockIdx.y * nslabs * slab_elems;
    C += (size_t)blockIdx.y * N * N;
    T3 += (size_t)blockIdx.y * nchunk * N * M2 * M;
    if (nchunk > 1) Cdup += (size_t)blockIdx.y * nchunk * N * N;

    // lane-invariant byte offsets inside a slab, relative to the (wave-uniform) start of the row
    // block 4i.  Only the k-step i_last = (N-1)/4 can straddle the end of the slab: its lanes with
    // row >= N get an out-of-range offset (the load is dropped and returns 0), and whole k-steps
    // beyond it (KCH > ceil(N/4)) are dropped through the scalar offset.
    constexpr int MINK = KCH == 4 ? 1 : KCH == 8 ? 5 : KCH == 11 ? 9 : KCH;   // smallest ceil(N/4) served
    const int i_last = (N - 1) / 4;
    const unsigned slab_bytes = (unsigned)(slab_elems * sizeof(double));
    const unsigned total_bytes = (unsigned)(nslabs * slab_elems * sizeof(double));
    const bool row_ok_last = 4 * i_last + lq < N;
    unsigned offp[NPA], offp_last[NPA];
    bool last_even[NPA];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int col = pp * 32 + 2 * lr;
        const int cc = col + 1 < N ? col : (N >= 2 ? N - 2 : 0);
        offp[pp] = (unsigned)((lq * N + cc) * sizeof(double));
        offp_last[pp] = row_ok_last ? offp[pp] : total_bytes;
        last_even[pp] = col == N - 1;
    }
    const int col1 = NP * 32 + lr;
    const unsigned offs = (unsigned)((lq * N + (col1 < N ? col1 : N - 1)) * sizeof(double));
    const unsigned offs_last = row_ok_last ? offs : total_bytes;

    // buffer loads: 32-bit per-lane byte offset (5 VGPRs for the whole slab) + the wave-uniform
    // slab / row-block offset in an SGPR, instead of one 64-bit VGPR address per load
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(g), 0, (int)total_bytes, 0x00020000);
    const unsigned rowblk_bytes = (unsigned)(4 * N * sizeof(double));
    auto soff = [&](unsigned sb, int i) -> unsigned {
        return (i < MINK || i <= i_last) ? sb + i * rowblk_bytes : total_bytes;
    };

    // this workgroup's task list and this wave's positions in it
    const int Tg = nchunk * N, W = gridDim.x, bx = blockIdx.x;
    const int ntask = (Tg - bx + W - 1) / W;
    const int npos = ntask * qc;
    if (nchunk > 1)   // rows of the duplicated coefficient matrix that belong to this workgroup's tasks
        for (int k = 0; k < ntask; ++k) {
            const int t = bx + k * W, p = t % N;
            for (int n = tid; n < N; n += HALF_WAVES * 64) Cdup[(size_t)t * N + n] = C[(size_t)p * N + n];
        }
    const int n_mine = wave < npos ? (npos - wave + HALF_WAVES - 1) / HALF_WAVES : 0;
    struct Pos { int k, qi; };
    auto advance = [&](Pos a) -> Pos {
        a.qi += HALF_WAVES;
        while (a.qi >= qc) { a.qi -= qc; ++a.k; }
        return a;
    };
    // byte offset of the slab at a position; positions past the task list or slots past q = N-1
    // become an out-of-range offset: the descriptor's range check drops those loads (no branch
    // here, so the s_waitcnt counts in the loop stay exact)
    auto slab_off = [&](Pos a) -> unsigned {
        const int t = bx + a.k * W;
        const int c = t / N, p = t - c * N, q = c * qc + a.qi;
        const bool ok = a.k < ntask && q < N;
        return __builtin_amdgcn_readfirstlane(ok ? (unsigned)(p * N + q) * slab_bytes : total_bytes);
    };
    auto issue = [&](Pos a, d2u (&ap)[NPA][KCH], double (&as)[KCH]) {
        const unsigned sb = slab_off(a);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc, (i >= MINK - 1 && i == i_last) ? offp_last[pp] : offp[pp], soff(sb, i), AUX_PLAIN);
                ap[pp][i] = __builtin_bit_cast(d2u, v);
            }
        if constexpr (NS1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(
                    rsrc, (i >= MINK - 1 && i == i_last) ? offs_last : offs, soff(sb, i), AUX_PLAIN);
                as[i] = __builtin_bit_cast(double, v);
            }
        }
    };
    // The loads of this wave's first slab go out before anything else: the LDS set-up and the C
    // fragment loads below (one L2 round trip and a barrier) then overlap the first HBM round trip.
    d2u ap0[NPA][KCH], ap1[NPA][KCH];
    double as0[KCH], as1[KCH];
    Pos p0{wave / qc, wave % qc};
    if (n_mine > 0) issue(p0, ap0, as0);

    for (int idx = tid; idx < CR * 16; idx += HALF_WAVES * 64) {
        const int r = idx >> 4, x = idx & 15;
        CxL[idx] = (r < N && x < M) ? C[(size_t)r * N + x] : 0.0;
    }
    for (int idx = tid; idx < nbuf * QR * ldb + 64; idx += HALF_WAVES * 64) blk[idx] = 0.0;
    if (tid < 5 * nbuf) cnt[tid] = (tid >= 2 * nbuf && tid < 3 * nbuf) ? (1 << 30) : 0;

    // stage-1 B fragments cfr[j] = C[4j + lq][lr]; stage-2 A fragments in MFMA row order:
    //   even/odd tile of pair pp: row m = lq + 4i <-> column pp*32 + 2(lq+4i) (+1)
    //   single tile             : column NP*32 + lq + 4i = row (NP*8 + i)*4 + lq  (= cfr[NP*8 + i])
    double cfr[NCF];
#pragma unroll
    for (int j = 0; j < NCF; ++j) {
        const int r = 4 * j + lq;
        cfr[j] = C[(size_t)(r < N ? r : N - 1) * N + (lr < M ? lr : M - 1)];
    }
    double cpr[NPA][2][4];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = pp * 32 + 2 * (lq + 4 * i) + half;
                cpr[pp][half][i] = C[(size_t)(col < N ? col : N - 1) * N + (lr < M ? lr : M - 1)];
            }
#pragma unroll
    for (int j = 0; j < NCF; ++j) cfr[j] *= ((4 * j + lq) < N && lr < M) ? 1.0 : 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cpr[pp][half][i] *= ((pp * 32 + 2 * (lq + 4 * i) + half) < N && lr < M) ? 1.0 : 0.0;
    __syncthreads();   // the only workgroup barrier

    // LDS destination of jt[i] = Jt[z = lq + 4i][y = lr] inside a block row: [y*M + z]
    int tile_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int zz = lq + 4 * i;
        tile_off[i] = (lr < M && zz < M) ? lr * M + zz : -1;
    }
    // One tile job of a complete block: stg[k][x][yz] = sum_slot C[q0+slot][x] blk[slot][yz] for the
    // 16 yz of tile ty (A = C rows: m = x, k = slot; B = block rows: k = slot, n = yz).  All
    // fragments are read before the MFMA chain (QR/4 <= NCF k-steps; the ones past QR/4 re-read
    // k-step 0 against a zero A operand).
    const int nty = (M2 + 15) / 16, kf = QR / 4;
    auto do_tile = [&](int k, int b, int ty) {
        const int t = bx + k * W;
        const int c = t / N;
        const double* arow = CxL + (size_t)(c * qc + lq) * 16 + lr;
        const double* brow = blk + ((size_t)b * QR + lq) * ldb + lr + 16 * ty;
        double af[NCF], bf[NCF];
#pragma unroll
        for (int i = 0; i < NCF; ++i) {
            const int ii = i < kf ? i : 0;
            af[i] = arow[(size_t)ii * 64];
            bf[i] = brow[(size_t)ii * 4 * ldb];
        }
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NCF; ++i) acc = mfma_f64(i < kf ? af[i] : 0.0, bf[i], acc);
        double* out = stg + (size_t)k * M * M2;
        const int yz = 16 * ty + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = lq + 4 * j;
            double* dst = (x < M && yz < M2) ? out + (size_t)x * M2 + yz : dump + lane;
            *dst = acc[j];
        }
    };
    // Take at most one tile job from the blocks that are open (complete, not yet consumed).  The
    // wave that finishes the last job of a block releases it (gen).  Returns whether a job was done.
    auto take_job = [&]() -> bool {
        for (int b = 0; b < nbuf; ++b) {
            if (__hip_atomic_load(&job[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= nty)
                continue;
            int j = 0;
            if (lane == 0)
                j = __hip_atomic_fetch_add(&job[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= nty) continue;
            asm volatile("" ::: "memory");
            const int k = __builtin_amdgcn_readfirstlane(
                __hip_atomic_load(&tsk[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            do_tile(k, b, j);
            asm volatile("" ::: "memory");
            int d = 0;
            if (lane == 0)
                d = __hip_atomic_fetch_add(&done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            d = __builtin_amdgcn_readfirstlane(d);
            if (d == nty - 1 && lane == 0)
                __hip_atomic_fetch_add(&gen[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return true;
        }
        return false;
    };
    auto compute = [&](Pos a, const d2u (&ap)[NPA][KCH], const double (&as)[KCH]) {
        d4 jt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                d4 xt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
                    // odd N: the lane whose even column is the LAST one loaded the pair
                    // (N-2, N-1), so its value sits in .y
                    const double ev = last_even[pp] ? ap[pp][i].y : ap[pp][i].x;
                    xt = mfma_f64(half == 0 ? ev : ap[pp][i].y, cfr[i], xt);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) jt = mfma_f64(cpr[pp][half][i], xt[i], jt);
            }
        if constexpr (NS1) {
            d4 xt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < KCH; ++i) xt = mfma_f64(as[i], cfr[i], xt);
#pragma unroll
            for (int i = 0; i < 4; ++i) jt = mfma_f64(cfr[NP * 8 + i], xt[i], jt);
        }
        if (a.k >= ntask) return;                     // padding position of the unrolled loop
        const int b = a.k % nbuf, need = a.k / nbuf;
        // The block is free once the previous task in it has been consumed.  If it is not (this wave
        // ran a whole ring ahead), help consuming: that rules out a deadlock (the oldest unconsumed
        // task can always complete: its arrivals are never blocked, and every waiting wave takes
        // its jobs).
        while (__hip_atomic_load(&gen[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
            if (!take_job()) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        double* row = blk + ((size_t)b * QR + a.qi) * ldb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = tile_off[i] >= 0 ? row + tile_off[i] : dump + lane;
            *dst = jt[i];
        }
        asm volatile("" ::: "memory");
        int old = 0;
        if (lane == 0)
            old = __hip_atomic_fetch_add(&cnt[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        asm volatile("" ::: "memory");
        if (old == qc - 1 && lane == 0) {             // block complete: open its tile jobs
            __hip_atomic_store(&cnt[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&tsk[b], a.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&done[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            __hip_atomic_store(&job[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("" ::: "memory");
        take_job();
    };

    if (n_mine > 0) {
        // Straight-line loop body (no exit test between an issue and its use: the optimiser sinks
        // loads below such a test, which serialises them with the MFMAs), and no VMEM store in it
        // (T3 is staged in LDS): the only VMEM traffic of the loop is the g_ao stream.  An odd
        // position count costs one idle MFMA pass on dropped loads.
        for (int it = 0; it < n_mine; it += 2) {
            const Pos p1 = advance(p0);
            issue(p1, ap1, as1);
            __builtin_amdgcn_sched_barrier(0);
            compute(p0, ap0, as0);
            __builtin_amdgcn_sched_barrier(0);
            const Pos p2 = advance(p1);
            issue(p2, ap0, as0);
            __builtin_amdgcn_sched_barrier(0);
            compute(p1, ap1, as1);
            __builtin_amdgcn_sched_barrier(0);
            p0 = p2;
        }
        while (take_job()) {}   // blocks opened by this wave's last arrivals
    }
    // every job has been run by a wave that is now past this point: the staged T3 is complete
    __syncthreads();
    for (int e = tid; e < ntask * M * M2; e += HALF_WAVES * 64) {
        const int k = e / (M * M2), idx = e - k * (M * M2);
        T3[(size_t)(bx + k * W) * M * M2 + idx] = stg[e];
    }
}

// Packed copy of p <-> q and r <-> s symmetric integrals for half_tri_kernel<.,.,2>: slab t = (p <= q)
// of the triangle; of each slab only the upper triangle, row r holding its columns (r & ~1) .. N-1
// (an even start keeps the 16-byte column-pair loads inside the row; the element left of the
// diagonal in an odd row is stored as 0) and the DIAGONAL HALVED, so that the slab product needs no
// weights: G = U + U^T with U = this triangle.  Rows back to back: row r starts
// 2k(N-k+1) + (r&1)(N-2k) doubles into the slab, k = r/2.  For N = 43: 968 of 1849 doubles per
// slab, 946 of 1849 slabs: 27 % of the tensor.  One workgroup per slab.
__device__ __host__ inline unsigned eri_tri_row_start(int r, int N)
{
    const int k = r >> 1;
    return (unsigned)(2 * k * (N - k + 1) + (r & 1) * (N - 2 * k));
}

// Pitch of a packed slab in doubles: the triangle rounded up to an even count, so that every slab
// (and every LDS slot it is copied to by 16-byte DMA lanes) starts on a 16-byte boundary.
__device__ __host__ inline unsigned eri_slab_pitch(int N) { return (eri_tri_row_start(N, N) + 1u) & ~1u; }

__global__ __launch_bounds__(256)
void eri_pack_kernel(const double* __restrict__ g, double* __restrict__ out, int N, unsigned slab_pk)
{
    const long t = blockIdx.x;
    const long tri = (long)N * (N + 1) / 2;
    int p, q;
    tri_decode(t, N, p, q);
    const double* src = g + (size_t)blockIdx.y * N * N * N * N + ((size_t)p * N + q) * N * N;
    double* dst = out + ((size_t)blockIdx.y * tri + t) * slab_pk;
    if (threadIdx.x == 0 && slab_pk > eri_tri_row_start(N, N)) dst[slab_pk - 1] = 0.0;   // the pad element
    for (int idx = threadIdx.x; idx < N * N; idx += 256) {
        const int r = idx / N, c = idx - r * N;
        const int e = r & ~1;
        if (c < e) continue;
        const double v = src[idx];
        dst[eri_tri_row_start(r, N) + (c - e)] = c < r ? 0.0 : (c == r ? 0.5 * v : v);
    }
}

// Ingest of a stack of AO tensors in ONE pass over them (N <= 48): the bitwise symmetry tests of
// eri_symmetry_check_kernel AND the packed copy of eri_pack_kernel.  One workgroup per (pair p <= q, geometry):
// slab (p,q) and its mirror (q,p) are read once each -- every slab of the tensor by exactly one workgroup --,
// compared word for word, slab (p,q) goes through LDS for the r <-> s test (the transposed element from LDS: the
// row length N is what it is, no 8-byte gathers from memory as in the two-pass check) and straight into the packed
// copy.  When the mirror differs, ITS r <-> s symmetry is tested as well (the flags are exact whatever the tensor).
// HBM traffic: the tensor read once + the copy written (27 % of it at N = 43); the two passes it replaces read the
// tensor 1.5 x for the check, the slabs p <= q again for the copy.  mismatch[g]: bit 0 = some slab differs from its
// mirror, bit 1 = some slab is not symmetric in (r, s).  `out` may be null (flags only).
__global__ __launch_bounds__(256)
void eri_ingest_kernel(const unsigned long long* __restrict__ g, double* __restrict__ out, int N, unsigned slab_pk,
                       int* __restrict__ mismatch)
{
    __shared__ unsigned long long sl[48 * 48];
    const long t = blockIdx.x;
    const long tri = (long)N * (N + 1) / 2;
    int p, q;
    tri_decode(t, N, p, q);
    const int n2 = N * N;
    const size_t geo = (size_t)blockIdx.y * n2 * n2;
    const unsigned long long* a = g + geo + ((size_t)p * N + q) * n2;
    const unsigned long long* b = g + geo + ((size_t)q * N + p) * n2;
    constexpr int PER = (48 * 48 + 255) / 256;          // 9 elements per thread at most
    unsigned long long va[PER];
    bool bad_pq = false;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = threadIdx.x + 256 * j;
        va[j] = idx < n2 ? a[idx] : 0ull;
    }
    if (p < q) {
        unsigned long long vb[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + 256 * j;
            vb[j] = idx < n2 ? b[idx] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) bad_pq |= va[j] != vb[j];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = threadIdx.x + 256 * j;
        if (idx < n2) sl[idx] = va[j];
    }
    const int any_pq = __syncthreads_or(bad_pq);
    bool bad_rs = false;
    double* dst = out ? out + ((size_t)blockIdx.y * tri + t) * slab_pk : nullptr;
    if (dst && threadIdx.x == 0 && slab_pk > eri_tri_row_start(N, N)) dst[slab_pk - 1] = 0.0;   // the pad element
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = threadIdx.x + 256 * j;
        if (idx >= n2) continue;
        const int r = idx / N, c = idx - r * N;
        if (r < c) bad_rs |= va[j] != sl[c * N + r];
        const int e = r & ~1;
        if (dst && c >= e) {
            const double v = __longlong_as_double((long long)va[j]);
            dst[eri_tri_row_start(r, N) + (c - e)] = c < r ? 0.0 : (c == r ? 0.5 * v : v);
        }
    }
    if (any_pq) {
        // the mirror slab is a different matrix: its own r <-> s symmetry
        __syncthreads();
        for (int idx = threadIdx.x; idx < n2; idx += 256) sl[idx] = b[idx];
        __syncthreads();
        for (int idx = threadIdx.x; idx < n2; idx += 256) {
            const int r = idx / N, c = idx - r * N;
            if (r < c) bad_rs |= sl[idx] != sl[c * N + r];
        }
    }
    if (bad_pq) atomicOr(mismatch + blockIdx.y, 1);
    if (bad_rs) atomicOr(mismatch + blockIdx.y, 2);
}

// ------------------------------------------------------------------------------------------
// Stage 1 over the upper triangle of slabs (p <-> q symmetric integrals; M <= 16, N <= 48),
// persistent and software-pipelined like half_transform_fused_kernel, whose load path it shares:
// wave gw of the W*8 waves of a geometry takes the slabs t = gw, gw + 8W, gw + 16W, ... of the
// row-major triangle (all waves stream neighbouring slabs at any moment), two slabs in flight in
// registers, buffer-descriptor loads with the range check as the only guard.  The 16x16 result
// tiles J[t][y][z] are STAGED IN LDS and written in one burst per phase (a phase = as many rounds
// as LDS holds; one phase for the batch shapes of the benchmark): a store stream interleaved with
// the read stream costs far more than its bytes.
// Algorithmic HBM bytes per geometry: 8 N^2 * N(N+1)/2 read + 8 M^2 * N(N+1)/2 written.
// ------------------------------------------------------------------------------------------
#ifdef OOVQE_TRI_PROBE
__device__ long long g_tri_wg_end[4096];
__device__ long long g_tri_cyc[16];      // workgroup (0,0): core cycles in the sweeps [wave], [8] the bursts, [9] all, [10] all in 100 MHz ticks
#endif

template <int KCH, int NST, int RS, int NWV = HALF_WAVES>
__global__ __launch_bounds__(NWV * 64)
void half_tri_kernel(const double* __restrict__ g, const double* __restrict__ C,
                     double* __restrict__ J, int N, int M, int phase_rounds, int tiled)
{
#ifdef OOVQE_TRI_PROBE
    const long long pc_start = __builtin_readcyclecounter();
    const long long pw_start = wall_clock64();
#endif
    constexpr int NP = NST / 2, NS1 = NST % 2;
    constexpr int NPA = NP > 0 ? NP : 1;
    constexpr int NCF = NST * 4;
    static_assert(KCH <= NCF, "C fragments must cover every k-step");
    static_assert(NP <= 1, "row blocks above a pair are not handled by the offsets below");
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int M2 = M * M;
    const long tri = (long)N * (N + 1) / 2;
    const size_t slab_elems = (size_t)N * N;
    double* dump = lds;             // [64] sink for lanes outside the M x M tile
    int* ctab = reinterpret_cast<int*>(lds + 64);   // [ncol <= 256] column -> position y*M + z in a tile
    double* stg = lds + 64 + 128;   // [phase_rounds][NWV][M2]
    // RS == 2: g is the PACKED copy made by eri_pack_kernel: per slab t = (p <= q) the upper triangle
    // with the diagonal halved, row r = its columns (r & ~1) .. N-1, rows back to back
    constexpr bool rs = RS != 0, pk = RS == 2;
    const unsigned slab_pk = (unsigned)(eri_slab_pitch(N) * sizeof(double));
#if defined(OOVQE_TRI_PROBE) && OOVQE_TRI_PROBE == 2
    // tools/tri_spread.hip: every geometry reads geometry 0's integrals (cache-resident): the kernel
    // without its HBM stream
#else
    g += (size_t)blockIdx.y * (pk ? (size_t)tri * (slab_pk / sizeof(double)) : slab_elems * slab_elems);
#endif
    C += (size_t)blockIdx.y * N * N;
    // tiled == 0: J[t][M2].  tiled == 1: J[ty][t][16], 16-wide tiles of the M2 (y z) columns
    // outermost (the layout sym_gm_kernel streams).  tiled == 2 (r <-> s symmetric integrals,
    // J[y][z] == J[z][y]): the same with the M(M+1)/2 columns y <= z only.
    const int ncol = tiled == 2 ? M * (M + 1) / 2 : M2;
    const int nty = (ncol + 15) / 16;
    J += (size_t)blockIdx.y * tri * (tiled ? nty * 16 : M2);
    if (tid < ncol) {               // read after the first barrier of the phase loop
        int y = tid / M, z = tid - y * M;
        if (tiled == 2) tri_decode(tid, M, y, z);
        ctab[tid] = (y * M + z) | ((z * M + y) << 16);   // position of (y,z) and of (z,y) in a tile
    }
    // r <-> s symmetric integrals (tiled == 2): a slab is itself symmetric, G = L + L^T + D with D
    // its diagonal 16x16 blocks and L the blocks above them.  Only D and L are LOADED (rows of a
    // lower block get an out-of-range offset: no traffic), the MFMA chain forms
    // T = C^T (L + D/2) C, and the burst writes J = T + T^T.  For N = 43: 67 % of the slab.
    // (RS != 0 goes with tiled == 2; a template parameter: the k-steps of lower blocks vanish at
    // compile time.  RS == 1 reads the full layout, where the rows keep their 8 N-byte pitch and
    // the skipped parts mostly share cache lines with the loaded ones: 7 % faster only; RS == 2
    // streams the packed copy.)
    // (the packed copy carries the weights itself: halved diagonal, nothing below it)
    constexpr bool wts = rs && !pk;
    const double wA_lo = wts ? 0.5 : 1.0, wB_lo = wts ? 0.0 : 1.0, wB_hi = wts ? 0.5 : 1.0, wD = wts ? 0.5 : 1.0;

    // Byte offsets inside a slab (see half_transform_fused_kernel): a per-lane part (VGPR) and a
    // wave-uniform part per k-step (SGPR).  Lanes that must not load (row >= N in the last k-step,
    // columns below the diagonal with rs) get an out-of-range offset: dropped, no traffic.
    //   full layout: row r = 4i + lq starts r * N doubles into the slab;
    //   packed     : row r starts 2k(N-k+1) + (r&1)(N-2k), k = r/2, and holds the columns >= (r & ~1);
    //                with r = 4i + lq, element (r, c) sits at  (4iN - 8i^2)  [uniform]
    //                + (2hN - 2h^2 + bN - 2hb) + c  [lane; h = lq>>1, b = lq&1]  - 4 i lq  [lane step per
    //                k-step]; the row's first column is r & ~1 = 4i + 2h.
    constexpr int MINK = KCH == 4 ? 1 : KCH == 8 ? 5 : KCH == 11 ? 9 : KCH;
    const int i_last = (N - 1) / 4;
    const unsigned slab_bytes = pk ? slab_pk : (unsigned)(slab_elems * sizeof(double));
    const unsigned total_bytes = pk ? (unsigned)(tri * slab_pk) : (unsigned)(slab_elems * slab_elems * sizeof(double));
    const bool row_ok_last = 4 * i_last + lq < N;
    const int h2 = lq >> 1, b2l = lq & 1;
    const int lane_row = pk ? 2 * h2 * N - 2 * h2 * h2 + b2l * N - 2 * h2 * b2l : lq * N;
    const unsigned lstep = pk ? (unsigned)(4 * lq * sizeof(double)) : 0u;   // subtracted once per k-step
    unsigned offp[NPA][2];              // pair pp: [0] rows of blocks <= 2pp, [1] rows of block 2pp+1 (full layout)
    int colp[NPA];
    bool last_even[NPA];
    // Packed copy: the two operands of pair pp are the columns of its first block (32 pp + lr, in .x)
    // and of its second block (32 pp + 16 + lr, in .y), one 8-byte load each: the first block's columns
    // only have rows up to their own block, so its products stop there (the column-pair form of the
    // full layout, .x = even and .y = odd columns from one 16-byte load, runs both over both blocks).
    int colh[NPA][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        if constexpr (pk) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = pp * 32 + 16 * h + lr;
                colh[pp][h] = col;
                offp[pp][h] = (unsigned)((lane_row + (col < N ? col : N - 1)) * (int)sizeof(double));
            }
            colp[pp] = colh[pp][0];
            last_even[pp] = false;
        } else {
            const int col = pp * 32 + 2 * lr;
            // the last column of an odd N: the pair (N-2, N-1) is loaded and .y taken
            const int cc = col + 1 < N ? col : (N >= 2 ? N - 2 : 0);
            colp[pp] = col;
            colh[pp][0] = colh[pp][1] = col;
            offp[pp][0] = (unsigned)((lane_row + cc) * (int)sizeof(double));
            offp[pp][1] = (wts && lr < 8) ? total_bytes : offp[pp][0];   // first block's columns below the diagonal
            last_even[pp] = col == N - 1;
        }
    }
    const int col1 = NP * 32 + lr;
    const int col1c = col1 < N ? col1 : N - 1;
    const unsigned offs = (unsigned)((lane_row + col1c) * (int)sizeof(double));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(g), 0, (int)total_bytes, 0x00020000);
    // scalar part of k-step i; k-steps past the last row are dropped
    auto soff = [&](unsigned sb, int i) -> unsigned {
        const unsigned o = pk ? sb + (unsigned)((4 * i * N - 8 * i * i) * (int)sizeof(double))
                              : sb + (unsigned)(4 * i * N * sizeof(double));
        return (i < MINK || i <= i_last) ? o : total_bytes;
    };
    // lane part of k-step i for a lane whose (first) column is col
    auto voff = [&](unsigned vo, int col, int i) -> unsigned {
        const bool drop = (i >= MINK - 1 && i == i_last && !row_ok_last) || (pk && col < 4 * i + 2 * h2);
        return drop ? total_bytes : vo - (unsigned)i * lstep;
    };

    const int SW = gridDim.x * NWV;                  // waves per geometry
    const int gw = blockIdx.x * NWV + wave;
    const int n_rounds = (int)((tri + SW - 1) / SW);        // the same for every wave of the grid
    // round r of this wave: slab t = r*SW + gw of the triangle; past the end -> dropped loads
    auto slab_off = [&](int r) -> unsigned {
        const long t = (long)r * SW + gw;
        if (pk) return __builtin_amdgcn_readfirstlane(t < tri ? (unsigned)t * slab_bytes : total_bytes);
        int p, q;
        tri_decode(t < tri ? t : 0, N, p, q);
        return __builtin_amdgcn_readfirstlane(t < tri ? (unsigned)(p * N + q) * slab_bytes : total_bytes);
    };
    auto issue = [&](int r, d2u (&ap)[NPA][KCH], double (&as)[KCH]) {
#if defined(OOVQE_TRI_PROBE) && OOVQE_TRI_PROBE == 3
        // tools/tri_spread.hip: the sweep without its loads (operands of the first slab for ever)
        if (r > 1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                asm volatile("" : "+v"(ap[0][i].x), "+v"(ap[0][i].y));
                asm volatile("" : "+v"(as[i]));
            }
            return;
        }
#endif
        const unsigned sb = slab_off(r);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const int blk = i / 4;                 // 16-row block of this k-step (compile time)
                if (rs && blk > 2 * pp + 1) continue;  // rows below both blocks of the pair: lower blocks
                if constexpr (pk) {
                    if (blk <= 2 * pp) {
                        const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff(offp[pp][0], colh[pp][0], i), soff(sb, i), AUX_NT);
                        ap[pp][i].x = __builtin_bit_cast(double, v);
                    }
                    const v2u w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff(offp[pp][1], colh[pp][1], i), soff(sb, i), AUX_NT);
                    ap[pp][i].y = __builtin_bit_cast(double, w);
                    continue;
                }
                // (row blocks above the pair, blk < 2pp, only exist for NST > 3: not instantiated)
                const unsigned vo = offp[pp][blk == 2 * pp + 1 ? 1 : 0];
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff(vo, colp[pp], i), soff(sb, i), rs ? AUX_NT : AUX_PLAIN);
                ap[pp][i] = __builtin_bit_cast(d2u, v);
            }
        if constexpr (NS1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff(offs, col1c, i), soff(sb, i), rs ? AUX_NT : AUX_PLAIN);
                as[i] = __builtin_bit_cast(double, v);
            }
        }
    };
    d2u ap0[NPA][KCH], ap1[NPA][KCH];
    double as0[KCH], as1[KCH];
    issue(0, ap0, as0);

    // C fragments: see half_transform_fused_kernel
    double cfr[NCF];
#pragma unroll
    for (int j = 0; j < NCF; ++j) {
        const int r = 4 * j + lq;
        cfr[j] = C[(size_t)(r < N ? r : N - 1) * N + (lr < M ? lr : M - 1)];
    }
    double cpr[NPA][2][4];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = pk ? pp * 32 + 16 * half + lq + 4 * i : pp * 32 + 2 * (lq + 4 * i) + half;
                cpr[pp][half][i] = C[(size_t)(col < N ? col : N - 1) * N + (lr < M ? lr : M - 1)];
            }
#pragma unroll
    for (int j = 0; j < NCF; ++j) cfr[j] *= ((4 * j + lq) < N && lr < M) ? 1.0 : 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cpr[pp][half][i] *= ((pk ? pp * 32 + 16 * half + lq + 4 * i : pp * 32 + 2 * (lq + 4 * i) + half) < N && lr < M) ? 1.0 : 0.0;

    // LDS destination of jt[i] = Jt[z = lq + 4i][y = lr] inside a staged tile: [y*M + z]
    int tile_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int zz = lq + 4 * i;
        tile_off[i] = (lr < M && zz < M) ? lr * M + zz : -1;
    }
    auto compute = [&](int r, int base, const d2u (&ap)[NPA][KCH], const double (&as)[KCH]) {
        d4 jt = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (pk) {
            // The packed copy carries its weights (halved diagonal): every first product is ONE
            // accumulator chain, its result goes to the second product as it leaves the MFMA pipe.
            // No VALU instruction in between: on gfx950 a VALU instruction does not run in the shadow
            // of a v_mfma_f64 (tools/mfma_dep_probe.hip: 8 cycles of the pipe each).
            d4 xs = d4{0.0, 0.0, 0.0, 0.0};
            if constexpr (NS1) {
#pragma unroll
                for (int i = 0; i < KCH; ++i) xs = mfma_f64(as[i], cfr[i], xs);
            }
            d4 xh[NPA][2];
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < KCH; ++i) {
                        if (i / 4 > 2 * pp + half) continue;   // rows below the block of these columns
                        x = mfma_f64(half == 0 ? ap[pp][i].x : ap[pp][i].y, cfr[i], x);
                    }
                    xh[pp][half] = x;
                }
            if constexpr (NS1) {
                // columns of the last tile that exist for this KCH (rows and columns share the k-steps)
                constexpr int KS1 = KCH - NP * 8 < 4 ? KCH - NP * 8 : 4;
#pragma unroll
                for (int i = 0; i < KS1; ++i) jt = mfma_f64(cfr[NP * 8 + i], xs[i], jt);
            }
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int i = 0; i < 4; ++i) jt = mfma_f64(cpr[pp][half][i], xh[pp][half][i], jt);
        } else {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // rows above the pair's blocks (xu), in its first block (xa), from its second on (xb)
                d4 xu = d4{0.0, 0.0, 0.0, 0.0}, xa = xu, xb = xu;
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
                    const double ev = last_even[pp] ? ap[pp][i].y : ap[pp][i].x;
                    const double av = half == 0 ? ev : ap[pp][i].y;
                    const int blk = i / 4;
                    if (rs && blk > 2 * pp + 1) continue;
                    if (blk < 2 * pp) xu = mfma_f64(av, cfr[i], xu);
                    else if (blk == 2 * pp) xa = mfma_f64(av, cfr[i], xa);
                    else xb = mfma_f64(av, cfr[i], xb);
                }
                // tile rows m = lq + 4e: e < 2 are columns of the first block (xa is their diagonal
                // block, xb lies below it), e >= 2 of the second (xa above, xb diagonal)
                d4 xt;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xt[e] = xu[e] + (e < 2 ? wA_lo : 1.0) * xa[e] + (e < 2 ? wB_lo : wB_hi) * xb[e];
#pragma unroll
                for (int i = 0; i < 4; ++i) jt = mfma_f64(cpr[pp][half][i], xt[i], jt);
            }
        if constexpr (NS1) {
            d4 xu = d4{0.0, 0.0, 0.0, 0.0}, xd = xu;   // rows above / inside the tile's diagonal block
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                if (i / 4 < NST - 1) xu = mfma_f64(as[i], cfr[i], xu);
                else xd = mfma_f64(as[i], cfr[i], xd);
            }
            d4 xt;
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[e] = xu[e] + wD * xd[e];
#pragma unroll
            for (int i = 0; i < 4; ++i) jt = mfma_f64(cfr[NP * 8 + i], xt[i], jt);
        }
        }
        if ((long)r * SW + gw >= tri) return;          // padding round
        double* row = stg + ((size_t)(r - base) * NWV + wave) * M2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = tile_off[i] >= 0 ? row + tile_off[i] : dump + lane;
            *dst = jt[i];
        }
    };

    // phase_rounds is even (double-buffered form): round `base` of every phase sits in register buffer
    // 0.  Straight-line loop body, no VMEM store inside it (see half_transform_fused_kernel).
    for (int base = 0; base < n_rounds; base += phase_rounds) {
        const int end = base + phase_rounds < n_rounds ? base + phase_rounds : n_rounds;
#ifdef OOVQE_TRI_PROBE
        const long long pc0 = __builtin_readcyclecounter();
#endif
        for (int it = base; it < end; it += 2) {
            issue(it + 1, ap1, as1);
            __builtin_amdgcn_sched_barrier(0);
            compute(it, base, ap0, as0);
            __builtin_amdgcn_sched_barrier(0);
            issue(it + 2, ap0, as0);
            __builtin_amdgcn_sched_barrier(0);
            compute(it + 1, base, ap1, as1);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef OOVQE_TRI_PROBE
        const long long pc1 = __builtin_readcyclecounter();
        if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && wave < 8) g_tri_cyc[wave] += pc1 - pc0;
#endif
        __syncthreads();
        // every wave writes the tiles of its own slabs (lanes = columns: 128-byte runs in the tiled
        // layouts), the rounds of the phase unrolled so that the LDS reads of several are in flight
        for (int c = lane; c < ncol; c += 64) {
            const int cc2 = ctab[c];
            const int o1 = cc2 & 0xffff, o2 = cc2 >> 16;
            double* dst = tiled ? J + ((size_t)(c >> 4) * tri) * 16 + (c & 15) : J + c;
            const size_t tstep = tiled ? 16 : (size_t)M2;
#pragma unroll 4
            for (int k = 0; k < end - base; ++k) {
                const long t = (long)(base + k) * SW + gw;
                if (t >= tri) break;
                const double* tile = stg + ((size_t)k * NWV + wave) * M2;
                double v = tile[o1];
                if (rs) v += tile[o2];                               // J = T + T^T
                dst[(size_t)t * tstep] = v;
            }
        }
        __syncthreads();
#ifdef OOVQE_TRI_PROBE
        if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_tri_cyc[8] += __builtin_readcyclecounter() - pc1;
#endif
    }
#ifdef OOVQE_TRI_PROBE
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        g_tri_cyc[9] += __builtin_readcyclecounter() - pc_start;
        g_tri_cyc[10] += wall_clock64() - pw_start;      // 100 MHz
    }
    // tools/half_standalone.hip: when does each workgroup finish (100 MHz wall clock)?
    if (tid == 0) g_tri_wg_end[blockIdx.y * gridDim.x + blockIdx.x] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------
// Stage 1 on the packed copy (both symmetries), slabs delivered by LDS-DMA.
//
// Same arithmetic and the same staging / burst of the results as half_tri_kernel<KCH,NST,2>; what
// changes is how a slab reaches the matrix cores.  There every lane fetches its MFMA operands
// straight from HBM: 8- and 16-byte pieces of ragged triangle rows, most of them straddling
// 128-byte lines, a third of the lanes masked off, and two slabs of operand registers per wave
// (234 VGPRs) to keep bytes in flight -- 4.7 TB/s where whole contiguous slabs stream at 5.7.
// Here a wave copies its packed slab (7.7 KB at N = 43, contiguous) into a private LDS slot with
// `global_load_lds_dwordx4` -- full 1 KB pieces, 16 bytes per lane, line-aligned, nothing masked
// but the tail of the last piece, no VGPR destination -- two slots per wave, so the slab after
// next is requested as soon as this one's operands have been read out of LDS into registers.
// The fragment reads then happen on chip (ds_read with the triangle's row offsets, zero for the
// positions below the diagonal) and cost ~60 LDS cycles per slab against the ~1000 cycles a CU
// has per slab at the HBM-bound pace.  Ordering: only the issuing wave reads its slots, so its
// own counted `s_waitcnt vmcnt` is the whole protocol (MI355X_MICROARCH.md, co-residence item 7);
// the burst's barriers are raw `s_barrier`s that leave the DMAs in flight.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void oovqe_lds_void;
typedef const __attribute__((address_space(1))) void oovqe_glob_void;

template <int KCH, int NST>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_tri_dma_kernel(const double* __restrict__ g, const double* __restrict__ C,
                         double* __restrict__ J, int N, int M, int phase_rounds)
{
    constexpr int NP = NST / 2, NS1 = NST % 2;
    constexpr int NPA = NP > 0 ? NP : 1;
    constexpr int NCF = NST * 4;
    static_assert(KCH <= NCF, "C fragments must cover every k-step");
    static_assert(NP <= 1, "row blocks above a pair are not handled by the offsets below");
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int M2 = M * M;
    const long tri = (long)N * (N + 1) / 2;
    const unsigned slab_d = eri_slab_pitch(N);                       // doubles per packed slab (even)
    const unsigned slab_bytes = slab_d * (unsigned)sizeof(double);
    const int npiece = (int)((slab_bytes + 1023u) / 1024u);          // 1 KB DMA pieces per slab
    double* dump = lds;             // [64] sink for lanes outside the M x M tile
    int* ctab = reinterpret_cast<int*>(lds + 64);   // [ncol <= 256] column -> position y*M + z in a tile
    const unsigned slot_d = slab_d + 2;                              // + a zero word the DMA never writes
    double* ring = lds + 64 + 128;  // [HALF_WAVES][2][slot_d]: two slots per wave
    double* stg = ring + (size_t)HALF_WAVES * 2 * slot_d;   // [phase_rounds][HALF_WAVES][M2]
    g += (size_t)blockIdx.y * (size_t)tri * slab_d;
    C += (size_t)blockIdx.y * N * N;
    const int ncol = M * (M + 1) / 2;                                // columns y <= z of J
    const int nty = (ncol + 15) / 16;
    J += (size_t)blockIdx.y * tri * nty * 16;                        // J[ty][t][16]
    if (tid < ncol) {               // read after the first barrier of the phase loop
        int y, z;
        tri_decode(tid, M, y, z);
        ctab[tid] = (y * M + z) | ((z * M + y) << 16);
    }

    const int SW = gridDim.x * HALF_WAVES;                  // waves per geometry
    const int gw = blockIdx.x * HALF_WAVES + wave;
    const int n_rounds = (int)((tri + SW - 1) / SW);        // the same for every wave of the grid
    const int my_rounds = gw < tri ? (int)((tri - 1 - gw) / SW) + 1 : 0;   // rounds with a slab for this wave
    double* my_ring = ring + (size_t)wave * 2 * slot_d;
    if (lane < 2) my_ring[(size_t)lane * slot_d + slab_d] = 0.0;
    // round r of this wave: slab t = r*SW + gw, into slot r & 1
    auto dma = [&](int r) {
        const char* src = reinterpret_cast<const char*>(g + ((size_t)r * SW + gw) * slab_d) + lane * 16;
        char* dst = reinterpret_cast<char*>(my_ring + (size_t)(r & 1) * slot_d);
        for (int p = 0; p < npiece; ++p) {
            if ((unsigned)(p * 1024 + lane * 16) < slab_bytes)
                __builtin_amdgcn_global_load_lds((oovqe_glob_void*)(src + p * 1024),
                                                 (oovqe_lds_void*)(dst + p * 1024), 16, 0, AUX_NT);
        }
    };
    if (my_rounds > 0) dma(0);
    if (my_rounds > 1) dma(1);

    // Offsets (in doubles) of this lane's operands inside a packed slab: row r = 4i + lq starts
    // 2k(N-k+1) + (r&1)(N-2k) doubles in, k = r/2, and holds the columns (r & ~1) .. N-1.
    // A position below the diagonal (or a row past N) reads as zero.
    // (elements outside the stored triangle point at the zero word behind the slot; pair pp = the
    // columns of its first block, k-steps of that block's rows only, and of its second block: see
    // half_tri_reg_kernel)
    const int zoff = (int)slab_d;
    int offp[NPA][2][KCH < 8 ? KCH : 8];
    int offs[KCH];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = pp * 32 + 16 * h + lr;
#pragma unroll
            for (int i = 0; i < (KCH < 8 ? KCH : 8); ++i) {
                const int r = 4 * i + lq, e = r & ~1;
                offp[pp][h][i] = (r < N && col >= e && col < N) ? (int)eri_tri_row_start(r, N) + col - e : zoff;
            }
        }
    {
        const int col1 = NP * 32 + lr;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int r = 4 * i + lq, e = r & ~1;
            offs[i] = (r < N && col1 < N && col1 >= e) ? (int)eri_tri_row_start(r, N) + col1 - e : zoff;
        }
    }

    // C fragments: see half_transform_fused_kernel
    double cfr[NCF];
#pragma unroll
    for (int j = 0; j < NCF; ++j) {
        const int r = 4 * j + lq;
        cfr[j] = C[(size_t)(r < N ? r : N - 1) * N + (lr < M ? lr : M - 1)];
    }
    double cpr[NPA][2][4];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = pp * 32 + 16 * half + lq + 4 * i;
                cpr[pp][half][i] = C[(size_t)(col < N ? col : N - 1) * N + (lr < M ? lr : M - 1)];
            }
#pragma unroll
    for (int j = 0; j < NCF; ++j) cfr[j] *= ((4 * j + lq) < N && lr < M) ? 1.0 : 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cpr[pp][half][i] *= ((pp * 32 + 16 * half + lq + 4 * i) < N && lr < M) ? 1.0 : 0.0;

    // LDS destination of jt[i] = Jt[z = lq + 4i][y = lr] inside a staged tile: [y*M + z]
    int tile_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int zz = lq + 4 * i;
        tile_off[i] = (lr < M && zz < M) ? lr * M + zz : -1;
    }

    double a0[NPA][4], a1[NPA][KCH < 8 ? KCH : 8];
    double as[KCH];
    // operands of the slab in slot `slot` out of LDS into registers (the slot is then free for the DMA
    // of the slab two rounds ahead)
    auto fetch = [&](int slot) {
        const double* sl = my_ring + (size_t)slot * slot_d;
        if constexpr (NS1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) as[i] = sl[offs[i]];
        }
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
            for (int i = 0; i < 4 && i < KCH; ++i) a0[pp][i] = sl[offp[pp][0][i]];
#pragma unroll
            for (int i = 0; i < (KCH < 8 ? KCH : 8); ++i) a1[pp][i] = sl[offp[pp][1][i]];
        }
    };
    // every first product one accumulator chain, its result straight into the second product
    auto compute = [&](int r, int base) {
        d4 jt = d4{0.0, 0.0, 0.0, 0.0};
        d4 xs = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (NS1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) xs = mfma_f64(as[i], cfr[i], xs);
        }
        d4 xh[NPA][2];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 4 && i < KCH; ++i) {
                if (i / 4 > 2 * pp) continue;
                x = mfma_f64(a0[pp][i], cfr[i], x);
            }
            xh[pp][0] = x;
            x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < (KCH < 8 ? KCH : 8); ++i) {
                if (i / 4 > 2 * pp + 1) continue;
                x = mfma_f64(a1[pp][i], cfr[i], x);
            }
            xh[pp][1] = x;
        }
        if constexpr (NS1) {
            constexpr int KS1 = KCH - NP * 8 < 4 ? KCH - NP * 8 : 4;
#pragma unroll
            for (int i = 0; i < KS1; ++i) jt = mfma_f64(cfr[NP * 8 + i], xs[i], jt);
        }
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < 4; ++i) jt = mfma_f64(cpr[pp][half][i], xh[pp][half][i], jt);
        double* row = stg + ((size_t)(r - base) * HALF_WAVES + wave) * M2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = tile_off[i] >= 0 ? row + tile_off[i] : dump + lane;
            *dst = jt[i];
        }
    };

    for (int base = 0; base < n_rounds; base += phase_rounds) {
        const int end = base + phase_rounds < n_rounds ? base + phase_rounds : n_rounds;
        for (int it = base; it < end; ++it) {
            if (it < my_rounds) {                         // wave-uniform
                // round `it` has landed once at most the pieces of round it+1 are outstanding
                // (vector-memory operations retire in issue order; the burst's stores count too
                // and only make this wait conservative)
                if (it + 1 < my_rounds) {
                    switch (npiece) {                     // wave-uniform; the count is an immediate
                    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                fetch(it & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // operands are in registers: the slot is free
                __builtin_amdgcn_sched_barrier(0);
                if (it + 2 < my_rounds) dma(it + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(it, base);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int nslot = (end - base) * HALF_WAVES;
        for (int e = tid; e < nslot * ncol; e += HALF_WAVES * 64) {
            const int slot = e / ncol, c = e - slot * ncol;
            const long t = (long)(base + slot / HALF_WAVES) * SW + blockIdx.x * HALF_WAVES + slot % HALF_WAVES;
            if (t < tri) {
                const int cc2 = ctab[c];
                const double v = stg[(size_t)slot * M2 + (cc2 & 0xffff)] + stg[(size_t)slot * M2 + (cc2 >> 16)];
                J[((size_t)(c >> 4) * tri + t) * 16 + (c & 15)] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

typedef __attribute__((address_space(3))) double lds_double_t;
__device__ __forceinline__ unsigned lds_address(const double* p)     // p points into LDS
{
    return (unsigned)(unsigned long)(lds_double_t*)p;
}

// ------------------------------------------------------------------------------------------
// Stage 1 on the packed copy, slabs fetched as CONTIGUOUS 16-byte-per-lane register loads.
//
// The packed stream of half_tri_kernel<.,.,2> runs at 4.7 TB/s where whole slabs reach 5.7, and
// neither the matrix cores nor the memory side is the reason: its operand-shaped loads leave a
// third of the lanes masked, so the two slabs a wave keeps in flight occupy 132 VGPRs but carry
// only 15 KB -- 124 KB per CU, against 236 KB for the whole-slab kernel (the LDS-DMA variant above
// has the same cap: its in-flight bytes ARE its LDS ring).  Here a slab is fetched as it lies in
// memory, NPC pieces of 1 KB per wave instruction, 4 VGPRs per piece: 32 VGPRs hold a 7.7 KB slab,
// so a wave keeps R = 3 or 4 slabs in flight (186 / 247 KB per CU) in the register budget the old
// kernel spends on two.  A landed slab is written to the wave's private LDS slot with
// ds_write_b128, its registers are reloaded with the slab R rounds ahead, and the MFMA operands
// are read back from LDS with the triangle's row offsets (zero below the diagonal) -- the same
// fragment reads as half_tri_dma_kernel.  Straight-line rounds, unrolled R times, so the
// compiler's vmcnt bookkeeping is exact; no barrier except around the result bursts.
// ------------------------------------------------------------------------------------------
template <int KCH, int NST, int NPC, int R>
__global__ __launch_bounds__(HALF_WAVES * 64)
void half_tri_reg_kernel(const double* __restrict__ g, const double* __restrict__ C,
                         double* __restrict__ J, int N, int M, int phase_rounds)
{
    constexpr int NP = NST / 2, NS1 = NST % 2;
    constexpr int NPA = NP > 0 ? NP : 1;
    constexpr int NCF = NST * 4;
    static_assert(KCH <= NCF, "C fragments must cover every k-step");
    static_assert(NP <= 1, "row blocks above a pair are not handled by the offsets below");
    extern __shared__ double lds[];
#ifdef OOVQE_TRI_PROBE
    const long long pc_start = __builtin_readcyclecounter();
    const long long pw_start = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int M2 = M * M;
    const long tri = (long)N * (N + 1) / 2;
    const unsigned slab_d = eri_slab_pitch(N);                       // doubles per packed slab (even)
    const unsigned slab_bytes = slab_d * (unsigned)sizeof(double);
    constexpr int SLOT_D = NPC * 128;                                // LDS slot of a wave, in doubles
    double* dump = lds;             // [64] sink for lanes outside the M x M tile
    int* ctab = reinterpret_cast<int*>(lds + 64);   // [ncol <= 256] column -> position y*M + z in a tile
    double* slots = lds + 64 + 128; // [HALF_WAVES][SLOT_D]
    double* stg = slots + (size_t)HALF_WAVES * SLOT_D;      // [phase_rounds][HALF_WAVES][M2]
#if defined(OOVQE_TRI_PROBE) && OOVQE_TRI_PROBE == 2
    // tools/tri_spread.hip: every geometry reads geometry 0's integrals (cache-resident)
#else
    g += (size_t)blockIdx.y * (size_t)tri * slab_d;
#endif
    C += (size_t)blockIdx.y * N * N;
    const int ncol = M * (M + 1) / 2;                                // columns y <= z of J
    const int nty = (ncol + 15) / 16;
    J += (size_t)blockIdx.y * tri * nty * 16;                        // J[ty][t][16]
    if (tid < ncol) {               // read after the first barrier of the phase loop
        int y, z;
        tri_decode(tid, M, y, z);
        ctab[tid] = (y * M + z) | ((z * M + y) << 16);
    }

    const int SW = gridDim.x * HALF_WAVES;                  // waves per geometry
    const int gw = blockIdx.x * HALF_WAVES + wave;
    const int n_rounds = (int)((tri + SW - 1) / SW);        // the same for every wave of the grid
    double* my_slot = slots + (size_t)wave * SLOT_D;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const unsigned total_bytes = (unsigned)(tri * slab_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(g), 0, (int)total_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)lane * 16u;
    // round r of this wave: slab t = r*SW + gw; past the end of the triangle -> out of range, dropped
    auto load = [&](int r, v4u (&b)[NPC]) {
#if defined(OOVQE_TRI_PROBE) && OOVQE_TRI_PROBE == 3
        if (r > 2) {                                    // tools/tri_spread.hip: the sweep without its loads
#pragma unroll
            for (int p = 0; p < NPC; ++p) asm volatile("" : "+v"(b[p]));
            return;
        }
#endif
        const int t = r * SW + gw;                      // (32-bit: scalar compare; tri <= 1 176)
        const unsigned sb = __builtin_amdgcn_readfirstlane(t < (int)tri ? (unsigned)t * slab_bytes : total_bytes);
#pragma unroll
        for (int p = 0; p < NPC; ++p)
            b[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + (unsigned)p * 1024u, sb, AUX_NT);
    };
    v4u buf[R][NPC];
#pragma unroll
    for (int b = 0; b < R; ++b) load(b, buf[b]);

    // Operand positions inside the wave's slot, in doubles relative to my_slot: pair pp = the columns
    // of its first block (32 pp + lr, k-steps of that block's rows only) and of its second block
    // (32 pp + 16 + lr); elements outside the stored triangle point at a zero word of LDS -- no select
    // between the ds_read and the MFMA (VALU instructions cost MFMA time on gfx950).
    double* zero_word = lds + 64 + 126;                     // (inside the unused tail of ctab's 256 ints)
    if (tid == 0) *zero_word = 0.0;
    const int zoff = (int)(zero_word - my_slot);
    int offp[NPA][2][KCH < 8 ? KCH : 8];
    int offs[KCH];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = pp * 32 + 16 * h + lr;
#pragma unroll
            for (int i = 0; i < (KCH < 8 ? KCH : 8); ++i) {
                const int r = 4 * i + lq, e = r & ~1;
                offp[pp][h][i] = (r < N && col >= e && col < N) ? (int)eri_tri_row_start(r, N) + col - e : zoff;
            }
        }
    {
        const int col1 = NP * 32 + lr;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int r = 4 * i + lq, e = r & ~1;
            offs[i] = (r < N && col1 < N && col1 >= e) ? (int)eri_tri_row_start(r, N) + col1 - e : zoff;
        }
    }

    double cfr[NCF];
#pragma unroll
    for (int j = 0; j < NCF; ++j) {
        const int r = 4 * j + lq;
        cfr[j] = C[(size_t)(r < N ? r : N - 1) * N + (lr < M ? lr : M - 1)];
    }
    double cpr[NPA][2][4];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = pp * 32 + 16 * half + lq + 4 * i;
                cpr[pp][half][i] = C[(size_t)(col < N ? col : N - 1) * N + (lr < M ? lr : M - 1)];
            }
#pragma unroll
    for (int j = 0; j < NCF; ++j) cfr[j] *= ((4 * j + lq) < N && lr < M) ? 1.0 : 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cpr[pp][half][i] *= ((pp * 32 + 16 * half + lq + 4 * i) < N && lr < M) ? 1.0 : 0.0;

    // LDS destination of jt[i] = Jt[z = lq + 4i][y = lr] inside a staged tile ([y*M + z]): the 32-bit LDS
    // address of the position in the phase's first row; lanes outside the M x M tile go to the dump
    unsigned st_mul[4], st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int zz = lq + 4 * i;
        const bool ok = lr < M && zz < M;
        st_mul[i] = ok ? 1u : 0u;
        st_off[i] = ok ? lds_address(stg) + (unsigned)((lr * M + zz) * (int)sizeof(double))
                       : lds_address(dump + lane);
    }

    // one round: slab in `b` -> LDS slot -> (registers reloaded R rounds ahead) -> fragments -> MFMA
    auto round = [&](int r, int base, v4u (&b)[NPC]) {
#pragma unroll
        for (int p = 0; p < NPC; ++p)
            *reinterpret_cast<v4u*>(reinterpret_cast<char*>(my_slot) + p * 1024 + lane * 16) = b[p];
        __builtin_amdgcn_sched_barrier(0);
        load(r + R, b);
        __builtin_amdgcn_sched_barrier(0);
        // every first product one accumulator chain, its result straight into the second product
        // (half_tri_kernel<.,.,2>); operands read from the slot as they are needed
        d4 jt = d4{0.0, 0.0, 0.0, 0.0};
        d4 xs = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (NS1) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) xs = mfma_f64(my_slot[offs[i]], cfr[i], xs);
        }
        d4 xh[NPA][2];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < (KCH < 8 ? KCH : 8); ++i) {
                    if (i / 4 > 2 * pp + half) continue;       // rows below the block of these columns
                    x = mfma_f64(my_slot[offp[pp][half][i]], cfr[i], x);
                }
                xh[pp][half] = x;
            }
        if constexpr (NS1) {
            constexpr int KS1 = KCH - NP * 8 < 4 ? KCH - NP * 8 : 4;
#pragma unroll
            for (int i = 0; i < KS1; ++i) jt = mfma_f64(cfr[NP * 8 + i], xs[i], jt);
        }
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < 4; ++i) jt = mfma_f64(cpr[pp][half][i], xh[pp][half][i], jt);
        if (r * SW + gw < (int)tri) {
            // byte address = row * (1 | 0) + (tile position | dump): one v_mad per store, no select
            const unsigned row_b = (unsigned)(((r - base) * HALF_WAVES + wave) * M2) * (unsigned)sizeof(double);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned a;        // (as asm: the optimiser turns the 0/1 product back into a select)
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a) : "s"(row_b), "v"(st_mul[i]), "v"(st_off[i]));
                *reinterpret_cast<lds_double_t*>((unsigned long)a) = jt[i];
            }
        }
        // the fragment reads of this round retire (lgkmcnt) before the next round's ds_writes to
        // the same slot are issued: same wave, LDS operations execute in order
    };

    __syncthreads();                                   // zero word, ctab
    // phase_rounds is a multiple of R: round `base` of every phase sits in register buffer 0
    for (int base = 0; base < n_rounds; base += phase_rounds) {
        const int end = base + phase_rounds < n_rounds ? base + phase_rounds : n_rounds;
        for (int it = base; it < end; it += R) {
#pragma unroll
            for (int b = 0; b < R; ++b) {
                round(it + b, base, buf[b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // loads stay in flight
        for (int c = lane; c < ncol; c += 64) {
            const int cc2 = ctab[c];
            const int o1 = cc2 & 0xffff, o2 = cc2 >> 16;
            double* dst = J + ((size_t)(c >> 4) * tri) * 16 + (c & 15);
#pragma unroll 4
            for (int kk = 0; kk < end - base; ++kk) {
                const long t = (long)(base + kk) * SW + gw;
                if (t >= tri) break;
                const double* tile = stg + ((size_t)kk * HALF_WAVES + wave) * M2;
                dst[(size_t)t * 16] = tile[o1] + tile[o2];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#ifdef OOVQE_TRI_PROBE
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        g_tri_cyc[9] += __builtin_readcyclecounter() - pc_start;
        g_tri_cyc[10] += wall_clock64() - pw_start;      // 100 MHz
    }
    if (tid == 0) g_tri_wg_end[blockIdx.y * gridDim.x + blockIdx.x] = wall_clock64();
#endif
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
                 double* c2, double* E, double* fock, double* gmat, double* gvec, double* dE,
                 const double* __restrict__ nuc_dev)
{
    extern __shared__ double lds[];
    if (nuc_dev) nuc = *nuc_dev;         // batched callers keep the nuclear repulsion on the device
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
//   in : U[n,q,y,z] = sum_p C[p,n] T2[p,q,y,z]   (from the K1 contraction kernel, T2 path)
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

    // Staging with the loads of up to 8 iterations in flight together (a plain load/store loop
    // pays one memory latency per iteration: 14 of them for U[n] at N = 43, M = 9)
    const double* Usrc = U + (size_t)n * N * M2;
    {
        const int total = N * M2;
        for (int base = 0; base < total; base += 8 * COL_THREADS) {
            double r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * COL_THREADS + tid;
                r[u] = idx < total ? Usrc[idx] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * COL_THREADS + tid;
                if (idx < total) Un[idx] = r[u];
            }
        }
        double rc[4], rn = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * COL_THREADS + tid, q = idx / M, x = idx - q * M;
            rc[u] = idx < N * M ? C[(size_t)q * N + x] : 0.0;
        }
        if (tid < N) rn = C[(size_t)tid * N + n];
        // first RDM chunk (two elements of Gamma per thread here, the rest in the loop)
        const int kc0 = nrdm < rdm_chunk ? nrdm : rdm_chunk;
        const double rg1 = tid < kc0 * na2 ? gamma[tid] : 0.0;
        const double rg2 = tid < kc0 * na4 ? Gamma[tid] : 0.0;
        const double rg3 = COL_THREADS + tid < kc0 * na4 ? Gamma[COL_THREADS + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * COL_THREADS + tid;
            if (idx < N * M) Cl[idx] = rc[u];
        }
        if (tid < N) cn[tid] = rn;
        if (tid < kc0 * na2) gml[tid] = rg1;
        if (tid < kc0 * na4) Gml[tid] = rg2;
        if (COL_THREADS + tid < kc0 * na4) Gml[COL_THREADS + tid] = rg3;
        for (int idx = COL_THREADS + tid; idx < kc0 * na2; idx += COL_THREADS) gml[idx] = gamma[idx];
        for (int idx = 2 * COL_THREADS + tid; idx < kc0 * na4; idx += COL_THREADS) Gml[idx] = Gamma[idx];
        for (int idx = 4 * COL_THREADS + tid; idx < N * M; idx += COL_THREADS) {
            const int q = idx / M, x = idx - q * M;
            Cl[idx] = C[(size_t)q * N + x];
        }
        for (int p = COL_THREADS + tid; p < N; p += COL_THREADS) cn[p] = C[(size_t)p * N + n];
    }
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
    // Gn[x,yz] = sum_q C[q,x] U[n,q,yz] on the MFMA: A = C^T from LDS (m = x, k = q), B = U[n] rows
    // from LDS (k = q, n = 16 consecutive yz); a wave owns the yz tiles wave, wave + 4, ...
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int lq = lane >> 4, lr = lane & 15;
        const int nyz = (M2 + 15) / 16, nxt = (M + 15) / 16, ksteps = (N + 3) / 4;
        for (int tile = wave; tile < nyz * nxt; tile += COL_THREADS / 64) {
            const int ty = tile % nyz, tx = tile / nyz;
            const int x = 16 * tx + lr, yz = 16 * ty + lr;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < ksteps; ++i) {
                const int q = 4 * i + lq;
                const int qc = q < N ? q : N - 1;
                const double av = Cl[qc * M + (x < M ? x : M - 1)] * ((q < N && x < M) ? 1.0 : 0.0);
                const double bv = Un[qc * M2 + (yz < M2 ? yz : M2 - 1)] * ((q < N && yz < M2) ? 1.0 : 0.0);
                acc = mfma_f64(av, bv, acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = 16 * tx + lq + 4 * j;
                if (xo < M && yz < M2) Gn[xo * M2 + yz] = acc[j];
            }
        }
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

    // RDM sets, rdm_chunk at a time through LDS.  gridDim.z workgroups share the chunks of one n (round 4: a
    // kUpCCD CAS(8e,8o) derivative evaluation carries 57 sets of 33 KB, one or two fit beside U[n]; with one
    // workgroup per n and one THREAD per (set, row) -- 16 threads busy, 512 serial terms each -- this loop took
    // 2.8 ms of a 3.2 ms Hessian).  Large active spaces: one WAVE per (set, row), the terms dealt to its lanes
    // and summed by a fixed shuffle tree.
    const int zsplit = gridDim.z, zid = blockIdx.z;
    const bool by_wave = na3 >= 128;
    const int lane = tid & 63, wave = tid >> 6;
    for (int k0 = zid * rdm_chunk; k0 < nrdm; k0 += zsplit * rdm_chunk) {
        const int kc = (nrdm - k0) < rdm_chunk ? (nrdm - k0) : rdm_chunk;
        __syncthreads();
        if (k0 > 0) {   // (chunk 0 was staged with the other inputs)
            for (int idx = tid; idx < kc * na2; idx += COL_THREADS) gml[idx] = gamma[(size_t)k0 * na2 + idx];
            for (int idx = tid; idx < kc * na4; idx += COL_THREADS) Gml[idx] = Gamma[(size_t)k0 * na4 + idx];
            __syncthreads();
        }
        if (by_wave) {
            // items: (set kl, row m) for the Fock columns, then (set kl) for the energy pieces
            for (int item = wave; item < kc * M + kc; item += COL_THREADS / 64) {
                double acc = 0.0;
                if (item < kc * M) {
                    const int kl = item / M, m = item - kl * M;
                    const double* gam = gml + (size_t)kl * na2;
                    if (m < no) {
                        for (int vw = lane; vw < na2; vw += 64) {
                            const int v = vw / na, w = vw - v * na, V = no + v, W = no + w;
                            acc += gam[vw] * (Gn[m * M2 + V * M + W] - 0.5 * Gn[W * M2 + V * M + m]);
                        }
                    } else {
                        const int v = m - no;
                        const double* Gv = Gml + (size_t)kl * na4 + (size_t)v * na3;
                        if (lane < na) acc = FIn[no + lane] * gam[v * na + lane];
                        for (int wxy = lane; wxy < na3; wxy += 64) {
                            const int w = wxy / na2, xy = wxy - w * na2, x = xy / na, y = xy - x * na;
                            acc += Gv[wxy] * Gn[(no + w) * M2 + (no + x) * M + no + y];
                        }
                    }
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                    if (lane == 0) {
                        const int k = k0 + kl;
                        Fcol[((size_t)k * M + m) * N + n] = m < no ? 2.0 * ((k == 0 ? FIn[m] : 0.0) + acc) : acc;
                    }
                } else {
                    const int kl = item - kc * M;
                    if (n >= no && n < M) {
                        const int p = n - no;
                        const double* gam = gml + (size_t)kl * na2 + (size_t)p * na;
                        const double* Gp = Gml + (size_t)kl * na4 + (size_t)p * na3;
                        if (lane < na) acc = FIn[no + lane] * gam[lane];
                        for (int qrs = lane; qrs < na3; qrs += 64) {
                            const int q = qrs / na2, rs = qrs - q * na2, r = rs / na, s2 = rs - r * na;
                            acc += 0.5 * Gn[(no + q) * M2 + (no + r) * M + no + s2] * Gp[qrs];
                        }
                        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                    }
                    if (lane == 0) Epart[(size_t)(k0 + kl) * N + n] = acc;
                }
            }
            continue;
        }
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

// ------------------------------------------------------------------------------------------
// Stage 3 on panels ("panel" kernel, T3 path): one workgroup per npan consecutive general indices
// n of one geometry, g_mo[n,x,y,z] supplied by the p -> n contraction (K1).  Same arithmetic, in
// the same order per n, as cas_column_kernel; npan is chosen so that a batched call fills the chip
// with one round of workgroups (a workgroup per n is 2752 small workgroups in ~1.5 rounds at 64
// geometries) and C[:, :M], h_ao and the RDM sets are staged once per panel instead of once per n.
// ------------------------------------------------------------------------------------------
constexpr int PAN_THREADS = 512;

__global__ __launch_bounds__(PAN_THREADS)
void cas_panel_kernel(const double* __restrict__ Gm_in, const double* __restrict__ h_ao,
                      const double* __restrict__ C, const double* __restrict__ gamma,
                      const double* __restrict__ Gamma, int nrdm, int N, int no, int na, int npan,
                      double* __restrict__ Fcol, double* __restrict__ Epart,
                      double* __restrict__ Cpart, double* __restrict__ c1, double* __restrict__ c2,
                      double* __restrict__ Gm_out, double* __restrict__ hmo_out, size_t out_stride,
                      int rdm_chunk, int batch_1d, const double* __restrict__ Wpre)
{
    extern __shared__ double lds[];
    const int M = no + na, M2 = M * M, M3 = M2 * M;
    const int na2 = na * na, na3 = na2 * na, na4 = na2 * na2;
    // Wpre [G][N][N]: W = C^T h_ao formed once per geometry by the launch of the circuit workgroups
    // (circuit.hip); h_ao and the panel's columns of C are then not staged here
    double* Gp = lds;                          // [npan][M3]   g_mo[n0+nl, x, y, z]
    double* Cl = Gp + (size_t)npan * M3;       // [N][M]       C[:, :M]
    double* hl = Cl + (size_t)N * M;           // [N][N]       h_ao                       (not with Wpre)
    double* cn = hl + (Wpre ? 0 : (size_t)N * N);           // [npan][N]    C[:, n]       (not with Wpre)
    double* Wn = cn + (Wpre ? 0 : (size_t)npan * N);        // [npan][N]    (C^T h)[n, :]
    double* hn = Wn + (size_t)npan * N;        // [npan][M]
    double* FIn = hn + (size_t)npan * M;       // [npan][M]
    double* gml = FIn + (size_t)npan * M;      // [rdm_chunk][na2]
    double* Gml = gml + (size_t)rdm_chunk * na2;   // [rdm_chunk][na4]
    const int tid = threadIdx.x;
    // 1-D grid (batch_1d > 0): index v = xcd + 8 (npanels k + panel) keeps the panels of geometry
    // xcd + 8 k on one XCD (see sym_gm_kernel): h_ao, C and the RDMs of a geometry enter one L2 only
    int bx = blockIdx.x, by = blockIdx.y;
    if (batch_1d > 0) {
        const int npanels = (N + npan - 1) / npan;
        const int v = blockIdx.x, sidx = v >> 3;
        bx = sidx % npanels;
        by = (v & 7) + 8 * (sidx / npanels);
        if (by >= batch_1d) return;
    }
    const int n0 = bx * npan;
    const int nn = (N - n0) < npan ? (N - n0) : npan;   // valid n in this panel
    {   // by = geometry of a batch: every per-geometry array is stacked
        const size_t gi = by;
        Gm_in += gi * (size_t)N * M3;
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
        if (Wpre) Wpre += gi * (size_t)N * N;
    }

    // every global load first (straight-line, into registers), then every LDS store: a load/store
    // loop per array would put one full memory latency per array in front of the barrier
    // (N <= 48, M <= 16, npan <= 16 bound the trip counts of the small arrays)
    {
        constexpr int IT_C = 2, IT_H = 5, IT_N = 2;
        double rc[IT_C], rh[IT_H], rn[IT_N];
#pragma unroll
        for (int it = 0; it < IT_C; ++it) {
            const int idx = tid + it * PAN_THREADS, q = idx / M, x = idx - q * M;
            rc[it] = idx < N * M ? C[(size_t)q * N + x] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < IT_H; ++it) {
            const int idx = tid + it * PAN_THREADS;
            rh[it] = (!Wpre && idx < N * N) ? h_ao[idx] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < IT_N; ++it) {
            const int idx = tid + it * PAN_THREADS, nl = idx / N, p = idx - nl * N;
            // (with Wpre: the panel's rows of W, contiguous)
            rn[it] = idx < nn * N ? (Wpre ? Wpre[(size_t)n0 * N + idx] : C[(size_t)p * N + n0 + nl]) : 0.0;
        }
        // first RDM chunk (one element of gamma and of Gamma per thread here, the rest in the loop)
        const int kc0 = nrdm < rdm_chunk ? nrdm : rdm_chunk;
        const double rg1 = tid < kc0 * na2 ? gamma[tid] : 0.0;
        const double rg2 = tid < kc0 * na4 ? Gamma[tid] : 0.0;
        // the panel of g_mo, eight loads in flight per thread
        {
            const double* src = Gm_in + (size_t)n0 * M3;
            const int total = nn * M3;
            for (int base = 0; base < total; base += 8 * PAN_THREADS) {
                double r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * PAN_THREADS + tid;
                    r[u] = idx < total ? src[idx] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * PAN_THREADS + tid;
                    if (idx < total) Gp[idx] = r[u];
                }
            }
        }
        if (tid < kc0 * na2) gml[tid] = rg1;
        if (tid < kc0 * na4) Gml[tid] = rg2;
        for (int idx = PAN_THREADS + tid; idx < kc0 * na2; idx += PAN_THREADS) gml[idx] = gamma[idx];
        for (int idx = PAN_THREADS + tid; idx < kc0 * na4; idx += PAN_THREADS) Gml[idx] = Gamma[idx];
#pragma unroll
        for (int it = 0; it < IT_C; ++it) {
            const int idx = tid + it * PAN_THREADS;
            if (idx < N * M) Cl[idx] = rc[it];
        }
#pragma unroll
        for (int it = 0; it < IT_H; ++it) {
            const int idx = tid + it * PAN_THREADS;
            if (!Wpre && idx < N * N) hl[idx] = rh[it];
        }
#pragma unroll
        for (int it = 0; it < IT_N; ++it) {
            const int idx = tid + it * PAN_THREADS;
            if (idx < nn * N) (Wpre ? Wn : cn)[idx] = rn[it];
        }
    }
    __syncthreads();
    // W[n,q] = sum_p C[p,n] h[p,q]
    for (int idx = tid; !Wpre && idx < nn * N; idx += PAN_THREADS) {
        const int nl = idx / N, q = idx - nl * N;
        const double* cv = cn + (size_t)nl * N;
        double a0 = 0.0, a1 = 0.0;
        int p = 0;
        for (; p + 1 < N; p += 2) {
            a0 += cv[p] * hl[(size_t)p * N + q];
            a1 += cv[p + 1] * hl[(size_t)(p + 1) * N + q];
        }
        if (p < N) a0 += cv[p] * hl[(size_t)p * N + q];
        Wn[idx] = a0 + a1;
    }
    __syncthreads();
    for (int idx = tid; idx < nn * M; idx += PAN_THREADS) {
        const int nl = idx / M, x = idx - nl * M;
        const double* Gn = Gp + (size_t)nl * M3;
        double acc = 0.0;
        for (int q = 0; q < N; ++q) acc += Wn[(size_t)nl * N + q] * Cl[q * M + x];
        hn[idx] = acc;
        double fi = acc;
        for (int i = 0; i < no; ++i) fi += 2.0 * Gn[x * M2 + i * M + i] - Gn[i * M2 + i * M + x];
        FIn[idx] = fi;
    }
    if (Gm_out && Gm_out != Gm_in)
        for (int idx = tid; idx < nn * M3; idx += PAN_THREADS) Gm_out[(size_t)n0 * M3 + idx] = Gp[idx];
    __syncthreads();
    if (hmo_out)
        for (int idx = tid; idx < nn * M; idx += PAN_THREADS) hmo_out[(size_t)n0 * M + idx] = hn[idx];

    // per-n pieces of the CAS coefficients (independent of the RDM sets)
    for (int nl = tid; nl < nn; nl += PAN_THREADS) {
        const int n = n0 + nl;
        Cpart[n] = n < no ? hn[nl * M + n] + FIn[nl * M + n] : 0.0;
    }
    for (int idx = tid; idx < nn * (na + na3); idx += PAN_THREADS) {
        const int nl = idx / (na + na3), r0 = idx - nl * (na + na3);
        const int n = n0 + nl;
        if (n < no || n >= M) continue;
        const int p = n - no;
        if (r0 < na) {
            c1[p * na + r0] = FIn[nl * M + no + r0];
        } else {
            int t = r0 - na;
            const int i3 = t;
            const int s = t % na; t /= na;
            const int r = t % na; t /= na;
            const int q = t;
            c2[(size_t)p * na3 + i3] = 0.5 * Gp[(size_t)nl * M3 + (no + q) * M2 + (no + r) * M + no + s];
        }
    }

    // RDM sets, rdm_chunk at a time through LDS
    for (int k0 = 0; k0 < nrdm; k0 += rdm_chunk) {
        const int kc = (nrdm - k0) < rdm_chunk ? (nrdm - k0) : rdm_chunk;
        __syncthreads();
        if (k0 > 0) {   // (chunk 0 was staged with the other inputs)
            for (int idx = tid; idx < kc * na2; idx += PAN_THREADS) gml[idx] = gamma[(size_t)k0 * na2 + idx];
            for (int idx = tid; idx < kc * na4; idx += PAN_THREADS) Gml[idx] = Gamma[(size_t)k0 * na4 + idx];
            __syncthreads();
        }
        // Fock columns: one thread per (n, set k, row m)
        for (int idx = tid; idx < nn * kc * M; idx += PAN_THREADS) {
            const int nl = idx / (kc * M), r0 = idx - nl * (kc * M);
            const int kl = r0 / M, m = r0 - kl * M, k = k0 + kl;
            const double* Gn = Gp + (size_t)nl * M3;
            const double* FIv = FIn + (size_t)nl * M;
            const double* gam = gml + (size_t)kl * na2;
            double val;
            if (m < no) {
                double fa = 0.0;
                for (int v = 0; v < na; ++v)
                    for (int w = 0; w < na; ++w) {
                        const int V = no + v, W = no + w;
                        fa += gam[v * na + w] * (Gn[m * M2 + V * M + W] - 0.5 * Gn[W * M2 + V * M + m]);
                    }
                val = 2.0 * ((k == 0 ? FIv[m] : 0.0) + fa);
            } else {
                const int v = m - no;
                const double* Gv = Gml + (size_t)kl * na4 + (size_t)v * na3;
                double acc = 0.0;
                for (int w = 0; w < na; ++w) acc += FIv[no + w] * gam[v * na + w];
                for (int w = 0; w < na; ++w)
                    for (int x = 0; x < na; ++x)
                        for (int y = 0; y < na; ++y)
                            acc += Gv[(w * na + x) * na + y] * Gn[(no + w) * M2 + (no + x) * M + no + y];
                val = acc;
            }
            Fcol[((size_t)k * M + m) * N + n0 + nl] = val;
        }
        // E_k contribution of row p = n - no (active n only), serial per (n, set): deterministic
        for (int idx = tid; idx < nn * kc; idx += PAN_THREADS) {
            const int nl = idx / kc, kl = idx - nl * kc;
            const int n = n0 + nl;
            double acc = 0.0;
            if (n >= no && n < M) {
                const int p = n - no;
                const double* Gn = Gp + (size_t)nl * M3;
                const double* FIv = FIn + (size_t)nl * M;
                const double* gam = gml + (size_t)kl * na2 + (size_t)p * na;
                const double* Gpq = Gml + (size_t)kl * na4 + (size_t)p * na3;
                for (int q = 0; q < na; ++q) acc += FIv[no + q] * gam[q];
                for (int q = 0; q < na; ++q)
                    for (int r = 0; r < na; ++r)
                        for (int s2 = 0; s2 < na; ++s2)
                            acc += 0.5 * Gn[(no + q) * M2 + (no + r) * M + no + s2]
                                   * Gpq[(q * na + r) * na + s2];
            }
            Epart[(size_t)(k0 + kl) * N + n] = acc;
        }
    }
}

// Stage 3 from MO integrals in memory, one workgroup (one wave) per (general index n, RDM set k):
// the same per-n arithmetic as cas_column_kernel (FI[n, :], column n of the generalized Fock
// matrix, the per-n pieces of c0 / c1 / c2 and of the energy) reading g_mo[n, x, y, z] from memory
// instead of LDS -- the staged path's M^3 block (140 KB at M = 26) does not fit there.  Replaces
// the one-workgroup fock_kernel inside oovqe_cas_eval: 481 -> ~15 us at N = 200, M = 26.
// cas_final_kernel assembles the outputs.
constexpr int FROW_THREADS = 256;
constexpr size_t FROW_LDS_MAX = 150 * 1024;

// doubles of LDS fock_rows_kernel needs for the slices of g_mo[n] it uses
static size_t fock_rows_lds_elems(int no, int na)
{
    const size_t M = (size_t)no + na;
    return 2 * M * no + 2 * (size_t)no * na * na + (size_t)na * na * na + 64 + 4;
}

__global__ __launch_bounds__(FROW_THREADS)
void fock_rows_kernel(const double* __restrict__ Gm, const double* __restrict__ hmo,
                      const double* __restrict__ gamma, const double* __restrict__ Gamma, int N, int no,
                      int na, double* __restrict__ Fcol, double* __restrict__ Epart,
                      double* __restrict__ Cpart, double* __restrict__ c1, double* __restrict__ c2)
{
    extern __shared__ double lds[];
    const int M = no + na, M2 = M * M, M3 = M2 * M;
    const int na2 = na * na, na3 = na2 * na, na4 = na2 * na2;
    const int n = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const double* gn = Gm + (size_t)n * M3;
    const double* gam = gamma + (size_t)k * na2;
    const double* Gam = Gamma + (size_t)k * na4;
    // the slices of g_mo[n] this index needs (2.7K of its 17.6K elements at M = 26), gathered by the
    // whole workgroup in one round trip: one wave walking them load by load is a chain of cache misses
    double* d1 = lds;                          // [M][no]      gn[x, i, i]
    double* d2 = d1 + (size_t)M * no;          // [M][no]      gn[i, i, x]
    double* o1 = d2 + (size_t)M * no;          // [no][na2]    gn[m, V, W]
    double* o2 = o1 + (size_t)no * na2;        // [no][na2]    gn[W, V, m]
    double* cube = o2 + (size_t)no * na2;      // [na3]        gn[act, act, act]
    double* FIn = cube + na3;                  // [64]
    double* red = FIn + 64;                    // [4]
    for (int idx = tid; idx < M * no; idx += FROW_THREADS) {
        const int x = idx / no, i = idx - x * no;
        d1[idx] = gn[x * M2 + i * M + i];
        d2[idx] = gn[i * M2 + i * M + x];
    }
    for (int idx = tid; idx < no * na2; idx += FROW_THREADS) {
        const int m = idx / na2, vw = idx - m * na2, v = vw / na, w = vw - v * na;
        o1[idx] = gn[m * M2 + (no + v) * M + no + w];
        o2[idx] = gn[(no + w) * M2 + (no + v) * M + m];
    }
    for (int idx = tid; idx < na3; idx += FROW_THREADS) {
        int t = idx;
        const int y = t % na; t /= na;
        const int x = t % na; t /= na;
        cube[idx] = gn[(no + t) * M2 + (no + x) * M + no + y];
    }
    __syncthreads();
    for (int x = tid; x < M; x += FROW_THREADS) {
        double fi = hmo[(size_t)n * M + x];
        for (int i = 0; i < no; ++i) fi += 2.0 * d1[x * no + i] - d2[x * no + i];
        FIn[x] = fi;
    }
    __syncthreads();
    for (int m = tid; m < M; m += FROW_THREADS) {
        double val;
        if (m < no) {
            double fa = 0.0;
            for (int vw = 0; vw < na2; ++vw) fa += gam[vw] * (o1[m * na2 + vw] - 0.5 * o2[m * na2 + vw]);
            val = 2.0 * ((k == 0 ? FIn[m] : 0.0) + fa);
        } else {
            const int v = m - no;
            const double* Gv = Gam + (size_t)v * na3;
            double acc = 0.0;
            for (int w = 0; w < na; ++w) acc += FIn[no + w] * gam[v * na + w];
            for (int idx = 0; idx < na3; ++idx) acc += Gv[idx] * cube[idx];
            val = acc;
        }
        Fcol[((size_t)k * M + m) * N + n] = val;
    }
    // E_k contribution of row p = n - no (active n only): fixed-order sum (lanes, then waves)
    double part = 0.0;
    if (n >= no && n < M) {
        const int p = n - no;
        const double* gp = gam + (size_t)p * na;
        const double* Gp = Gam + (size_t)p * na3;
        for (int q = tid; q < na; q += FROW_THREADS) part += FIn[no + q] * gp[q];
        for (int idx = tid; idx < na3; idx += FROW_THREADS) part += 0.5 * cube[idx] * Gp[idx];
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) Epart[(size_t)k * N + n] = (red[0] + red[1]) + (red[2] + red[3]);
    if (k == 0) {
        if (tid == 0) Cpart[n] = n < no ? hmo[(size_t)n * M + n] + FIn[n] : 0.0;
        if (n >= no && n < M) {
            const int p = n - no;
            for (int q = tid; q < na; q += FROW_THREADS) c1[p * na + q] = FIn[no + q];
            for (int idx = tid; idx < na3; idx += FROW_THREADS) c2[(size_t)p * na3 + idx] = 0.5 * cube[idx];
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
    // four entries per thread at a time: index loads, then the dependent Fock loads, then the
    // stores (a plain loop pays two memory latencies per entry)
    for (long base = 0; base < (long)nrdm * n_kappa; base += 4 * 512) {
        int rr[4], cc[4], kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long idx = base + u * 512 + tid;
            const bool ok = idx < (long)nrdm * n_kappa;
            const long ii = ok ? idx : 0;
            kk[u] = (int)(ii / n_kappa);
            const int t = (int)(ii - (long)kk[u] * n_kappa);
            rr[u] = kap_row[t];
            cc[u] = kap_col[t];
        }
        double frc[4], fcr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* F = Fcol + (size_t)kk[u] * M * N;
            const int r = rr[u], c = cc[u];
            frc[u] = F[(size_t)(r < M ? r : 0) * N + c] * (r < M ? 1.0 : 0.0);
            fcr[u] = F[(size_t)(c < M ? c : 0) * N + r] * (c < M ? 1.0 : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long idx = base + u * 512 + tid;
            if (idx < (long)nrdm * n_kappa) gvec[idx] = 2.0 * (frc[u] - fcr[u]);
        }
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
                                  int batch, oovqe_stream_t stream, int sym = SYM_FULL, bool rs = false,
                                  double* Vk_tri = nullptr, const double* g_tiles = nullptr);
static int device_cu_count();

extern "C" int oovqe_cas_half_transform(const double* g_ao, const double* C, int N, int M, double* T2,
                                        oovqe_stream_t stream)
{
    return half_transform_batched(g_ao, C, N, M, T2, 1, stream);
}

static int half_transform_batched(const double* g_ao, const double* C, int N, int M, double* T2,
                                  int batch, oovqe_stream_t stream, int sym, bool rs, double* Vk_tri,
                                  const double* g_tiles)
{
    // g_tiles: the tile-packed copy of g_ao (oovqe_eri_pack, N > 48; both symmetry flags) or null
    OOVQE_REQUIRE(g_ao && C && T2, "cas_half_transform: null pointer");
    OOVQE_REQUIRE(!Vk_tri || (sym == SYM_MIRROR && N <= 48),
                  "cas_half_transform: the quarter-transformed output needs p<->q symmetric integrals and N <= 48");
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
    const long nslabs = sym == SYM_FULL ? (long)N * N : (long)N * (N + 1) / 2;
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
        oovqe_note_stage1("half_transform_kernel<%d,%d,%d>%s", Z, KC_, NS_, sym ? " (slabs p <= q)" : ""); \
        const unsigned wpg = (NS_) > 0 ? 4u : (unsigned)HALF_WAVES;       /* waves per workgroup */      \
        hipLaunchKernelGGL((half_transform_kernel<Z, KC_, NS_>),                                  \
                           dim3((unsigned)((nslabs + wpg - 1) / wpg), batch), dim3(wpg * 64),     \
                           (NS_) > 0 ? 0 : lds_bytes, st, g_ao, C, T2, N, M, nrb, nkc, nslabs, sym, Vk_tri); \
    } while (0)
#define OOVQE_DISPATCH_KCH(Z)                                                                     \
    do {                                                                                          \
        /* whole-slab preload when the slab is one k-chunk deep and <= 48 loads per lane */      \
        if (kch == 4 && nrb == 1) OOVQE_LAUNCH_HALF(Z, 4, 1);                                     \
        else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_HALF(Z, 8, 2);                                \
        else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_HALF(Z, 11, 3);                              \
        else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_HALF(Z, 12, 3);                              \
        else OOVQE_LAUNCH_HALF(Z, 16, 0);   /* N > 48 (kch == 16 there) */                        \
    } while (0)
    // N > 48: persistent streaming kernel.  k-chunk depth with the least padding (ties: deeper)
    const bool stream_old = oovqe_opt(OOVQE_OPT_HALF_STREAM_OLD) != 0;   // A/B hook for tools/
    if (nrb > 3 && !stream_old) {
        int skch = 16, waste = 1 << 30;
        for (int k = 16; k >= 10; --k) {
            const int w = (ksteps + k - 1) / k * k - ksteps;
            if (w < waste) { waste = w; skch = k; }
        }
        const int snkc = (ksteps + skch - 1) / skch;
        const long want = (nslabs + HALF_WAVES - 1) / HALF_WAVES;
        // r <-> s symmetric slabs: upper tile triangle only (needs a 16 x 17 patch per wave in LDS)
        const size_t lds_rs = lds_bytes + (size_t)HALF_WAVES * 16 * 17 * sizeof(double);
        const bool use_rs = rs && lds_rs <= 160 * 1024;
        if (g_tiles && use_rs && sym == SYM_MIRROR) {
            // both flags and a resident tile-packed copy: half_tiles_kernel streams that instead
            const long slab_doubles = (long)nrb * (nrb + 1) / 2 * 256;
#define OOVQE_LAUNCH_TILES(Z, D_, A_)                                                             \
    do {                                                                                          \
        const void* fn = (const void*)half_tiles_kernel<Z, D_, A_>;                               \
        int rc_ = oovqe_ensure_dynamic_lds(fn, 160 * 1024);                                       \
        if (rc_) return rc_;                                                                      \
        long per = (long)device_cu_count() / batch;      /* one workgroup per CU (LDS) */         \
        if (per < 1) per = 1;                                                                     \
        if (per > want) per = want;                                                               \
        oovqe_note_stage1("half_tiles_kernel<%d,%d>", Z, D_);                                     \
        hipLaunchKernelGGL((half_tiles_kernel<Z, D_, A_>), dim3((unsigned)per, batch),            \
                           dim3(HALF_WAVES * 64), lds_rs, st, g_tiles, C, T2, N, M, nrb, nslabs,  \
                           nslabs * slab_doubles);                                                \
    } while (0)
            const int variant = oovqe_opt(OOVQE_OPT_TILES_VARIANT);
            oovqe_profile_mark_start(st);
            if (ZT == 1) OOVQE_LAUNCH_TILES(1, 8, AUX_NT);
            else if (ZT == 3) OOVQE_LAUNCH_TILES(3, 4, AUX_NT);
            else if (variant == 1) OOVQE_LAUNCH_TILES(2, 4, AUX_NT);
            else if (variant == 2) OOVQE_LAUNCH_TILES(2, 6, AUX_NT);
            else if (variant == 4) OOVQE_LAUNCH_TILES(2, 8, AUX_PLAIN);
            else OOVQE_LAUNCH_TILES(2, 8, AUX_NT);   // (N = 200, M = 26: 833 us per evaluation; ring of 6: 860-920, of 4: 858; default cache policy: 900)
            oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_TILES
            OOVQE_CHECK_LAUNCH("cas_half_transform/tiles");
            return 0;
        }
#define OOVQE_LAUNCH_STREAM(Z, KC_, D_)                                                           \
    do {                                                                                          \
        if (use_rs) OOVQE_LAUNCH_STREAM_RS(Z, KC_, D_, true, lds_rs);                             \
        else OOVQE_LAUNCH_STREAM_RS(Z, KC_, D_, false, lds_bytes);                                \
    } while (0)
#define OOVQE_LAUNCH_STREAM_RS(Z, KC_, D_, RS_, lds_bytes)                                        \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        static size_t occ_lds = 0;                                                                \
        static int occ = 1;                                                                       \
        const void* fn = (const void*)half_stream_kernel<Z, KC_, D_, RS_>;                        \
        if (!attr_done) {                                                                         \
            OOVQE_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                                160 * 1024), "cas_half_transform");               \
            attr_done = true;                                                                     \
        }                                                                                         \
        if (occ_lds != lds_bytes) {                                                               \
            int nb = 0;                                                                           \
            OOVQE_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, HALF_WAVES * 64,\
                                                                         lds_bytes),              \
                            "cas_half_transform");                                                \
            occ = nb < 1 ? 1 : nb;                                                                \
            occ_lds = lds_bytes;                                                                  \
        }                                                                                         \
        long per = ((long)occ * device_cu_count()) / batch;                                       \
        if (per < 1) per = 1;                                                                     \
        if (per > want) per = want;                                                               \
        oovqe_note_stage1("half_stream_kernel<%d,%d,%d,%s>", Z, KC_, D_, RS_ ? "true" : "false"); \
        hipLaunchKernelGGL((half_stream_kernel<Z, KC_, D_, RS_>), dim3((unsigned)per, batch),     \
                           dim3(HALF_WAVES * 64), lds_bytes, st, g_ao, C, T2, N, M, nrb, snkc,    \
                           nslabs, sym);                                                          \
    } while (0)
#define OOVQE_DISPATCH_STREAM(Z, D_)                                                              \
    do {                                                                                          \
        switch (skch) {                                                                           \
        case 10: OOVQE_LAUNCH_STREAM(Z, 10, D_); break;                                           \
        case 11: OOVQE_LAUNCH_STREAM(Z, 11, D_); break;                                           \
        case 12: OOVQE_LAUNCH_STREAM(Z, 12, D_); break;                                           \
        case 13: OOVQE_LAUNCH_STREAM(Z, 13, D_); break;                                           \
        case 14: OOVQE_LAUNCH_STREAM(Z, 14, D_); break;                                           \
        case 15: OOVQE_LAUNCH_STREAM(Z, 15, D_); break;                                           \
        default: OOVQE_LAUNCH_STREAM(Z, 16, D_); break;                                           \
        }                                                                                         \
    } while (0)
        oovqe_profile_mark_start(st);
        if (ZT == 1) OOVQE_DISPATCH_STREAM(1, 3);
        else if (ZT == 2) OOVQE_DISPATCH_STREAM(2, 3);
        else OOVQE_DISPATCH_STREAM(3, 2);
        oovqe_profile_mark_stop(st);
#undef OOVQE_DISPATCH_STREAM
#undef OOVQE_LAUNCH_STREAM
#undef OOVQE_LAUNCH_STREAM_RS
        OOVQE_CHECK_LAUNCH("cas_half_transform");
        return 0;
    }
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

extern "C" int oovqe_eri_symmetry_flags(const double* g_ao, int N, int batch, unsigned* eri_flags,
                                        oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && eri_flags, "eri_symmetry_flags: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 65535 && batch >= 1 && batch <= 65535, "eri_symmetry_flags: N=%d batch=%d",
                  N, batch);
    hipStream_t st = (hipStream_t)stream;
    int* flag = nullptr;
    OOVQE_CHECK_HIP(hipMalloc(&flag, sizeof(int)), "eri_symmetry_flags");
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(eri_symmetry_check_kernel, dim3(N, N, batch), dim3(256), 0, st,
                           reinterpret_cast<const unsigned long long*>(g_ao), N, flag);
        e = hipGetLastError();
    }
    int bad = 3;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(flag);
    OOVQE_CHECK_HIP(e, "eri_symmetry_flags");
    *eri_flags = ((bad & 1) ? 0u : OOVQE_ERI_PQ_SYMMETRIC) | ((bad & 2) ? 0u : OOVQE_ERI_RS_SYMMETRIC);
    return 0;
}

// Ingest of a stack of AO tensors: per-geometry symmetry flags (host array [batch]) and, when `packed` is given,
// the packed resident copy of EVERY geometry (meaningful for those whose flags carry both bits) -- one pass over
// g_ao for N <= 48 (eri_ingest_kernel), the check pass + the tile pack beyond.  Synchronises `stream`.
extern "C" int oovqe_eri_ingest(const double* g_ao, int N, int batch, double* packed, unsigned* eri_flags,
                                oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && eri_flags, "eri_ingest: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 4096 && batch >= 1 && batch <= 65535, "eri_ingest: N=%d batch=%d", N, batch);
    hipStream_t st = (hipStream_t)stream;
    if (N > 48) {
        for (int gi = 0; gi < batch; ++gi) {
            int rc = oovqe_eri_symmetry_flags(g_ao + (size_t)gi * N * N * N * N, N, 1, eri_flags + gi, stream);
            if (rc) return rc;
        }
        return packed ? oovqe_eri_pack(g_ao, N, batch, packed, stream) : 0;
    }
    int* flag = nullptr;
    OOVQE_CHECK_HIP(hipMalloc(&flag, sizeof(int) * (size_t)batch), "eri_ingest");
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int) * (size_t)batch, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(eri_ingest_kernel, dim3((unsigned)(N * (N + 1) / 2), batch), dim3(256), 0, st,
                           reinterpret_cast<const unsigned long long*>(g_ao), packed, N, eri_slab_pitch(N), flag);
        e = hipGetLastError();
    }
    int* bad = (int*)malloc(sizeof(int) * (size_t)batch);
    if (!bad) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpyAsync(bad, flag, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(flag);
    if (e == hipSuccess)
        for (int gi = 0; gi < batch; ++gi)
            eri_flags[gi] = ((bad[gi] & 1) ? 0u : OOVQE_ERI_PQ_SYMMETRIC) | ((bad[gi] & 2) ? 0u : OOVQE_ERI_RS_SYMMETRIC);
    free(bad);
    OOVQE_CHECK_HIP(e, "eri_ingest");
    return 0;
}

static int sym_q_contract_batched(const double* J, const double* C, double* T3, int N, int M, int batch,
                                  hipStream_t st)
{
    OOVQE_REQUIRE(M >= 1 && M <= 16 && N >= 1 && N <= 48, "sym_q_contract: N=%d M=%d", N, M);
    const unsigned threads = (unsigned)((M * M + 15) / 16 * 64);
    const int ksteps = (N + 3) / 4;
    oovqe_profile_mark_start_l(st, 5);
    if (ksteps <= 4)
        hipLaunchKernelGGL(sym_q_contract_kernel<4>, dim3(N, batch), dim3(threads), 0, st, J, C, T3, N, M);
    else if (ksteps <= 8)
        hipLaunchKernelGGL(sym_q_contract_kernel<8>, dim3(N, batch), dim3(threads), 0, st, J, C, T3, N, M);
    else
        hipLaunchKernelGGL(sym_q_contract_kernel<12>, dim3(N, batch), dim3(threads), 0, st, J, C, T3, N, M);
    oovqe_profile_mark_stop(st);
    OOVQE_CHECK_LAUNCH("cas_eval/sym_q_contract");
    return 0;
}

// Gm [G][N][M^3] from the packed triangle J in one launch (+ the circuit workgroups of cj, if any)
static int sym_gm_batched(const double* J, const double* C, double* Gm, int N, int M, int batch,
                          hipStream_t st, const oovqe_circuit_job_t* cj, bool packed)
{
    OOVQE_REQUIRE(M >= 1 && M <= 16 && N >= 1 && N <= 48, "sym_gm: N=%d M=%d", N, M);
    const int ksteps = (N + 3) / 4;
    int KSr = ksteps <= 4 ? 4 : ksteps <= 8 ? 8 : 12;
    size_t lds_bytes = (size_t)4 * KSr * (M * 16 + 8) * sizeof(double);
    if (cj && cj->lds_bytes > lds_bytes) lds_bytes = cj->lds_bytes;
    // N = 41 ... 44 (cc-pVDZ formaldimine: 43): T3 in exactly 44 rows leaves room for a THIRD workgroup per CU (3 x
    // 53.5 KB), built for 80 VGPRs with one row of J in flight per wave.  Measured (round 4, 256 geometries): 86 us
    // against 58 us for two workgroups per CU with two rows in flight -- 24 registers spilled and half the loads in
    // flight per wave cost more than the third workgroup hides.  Kept behind the option gm_three_per_cu = 1.
    const size_t lds11 = (size_t)44 * (M * 16 + 8) * sizeof(double);
    const bool three_per_cu = ksteps == 11 && (!cj || cj->lds_bytes <= lds11) && 3 * lds11 <= 160 * 1024 &&
                              oovqe_opt(OOVQE_OPT_GM_THREE_PER_CU) == 1;
    const unsigned nty = (unsigned)(((packed ? M * (M + 1) / 2 : M * M) + 15) / 16);
    oovqe_circuit_job_t job;
    memset(&job, 0, sizeof(job));
    if (cj) job = *cj;
    const unsigned nx = nty + (cj ? 1 : 0);
    const bool xcd_grid = batch > 1 && oovqe_opt(OOVQE_OPT_GM_PLAIN_GRID) == 0;
    const dim3 grid = xcd_grid ? dim3(nx * (unsigned)((batch + 7) / 8 * 8)) : dim3(nx, batch);
#define OOVQE_LAUNCH_GM2(KS_, WPE_)                                                               \
    do {                                                                                          \
        static size_t attr_bytes = 0;                                                             \
        if (lds_bytes > attr_bytes) {                                                             \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)sym_gm_kernel<KS_, WPE_>,            \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                                (int)lds_bytes), "cas_eval/sym_gm");              \
            attr_bytes = lds_bytes;                                                               \
        }                                                                                         \
        hipLaunchKernelGGL((sym_gm_kernel<KS_, WPE_>), grid, dim3(SMALL_THREADS), lds_bytes, st,  \
                           J, C, Gm, N, M, packed ? 1 : 0, job, cj ? 1 : 0, batch);               \
    } while (0)
    // (measurement hooks: OOVQE_GM_TWO_PER_CU / OOVQE_GM_ONE_PER_CU force a build; 128 geometries:
    // 45 -> 38 us with two per CU, 64 geometries: no difference)
    const bool multi_round = ((long)(nty + (cj ? 1 : 0)) * batch > (long)device_cu_count() || oovqe_opt(OOVQE_OPT_GM_TWO_PER_CU)) &&
                             2 * lds_bytes <= 160 * 1024 && oovqe_opt(OOVQE_OPT_GM_ONE_PER_CU) == 0;
#define OOVQE_LAUNCH_GM(KS_)                                                                      \
    do {                                                                                          \
        if (multi_round) OOVQE_LAUNCH_GM2(KS_, 4);                                                \
        else OOVQE_LAUNCH_GM2(KS_, 2);                                                            \
    } while (0)
    oovqe_profile_mark_start_l(st, 2);
    if (three_per_cu) {
        KSr = 11;
        lds_bytes = lds11;
        OOVQE_LAUNCH_GM2(11, 6);
    } else if (KSr == 4) OOVQE_LAUNCH_GM(4);
    else if (KSr == 8) OOVQE_LAUNCH_GM(8);
    else OOVQE_LAUNCH_GM(12);
    oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_GM
#undef OOVQE_LAUNCH_GM2
    OOVQE_CHECK_LAUNCH("cas_eval/sym_gm");
    return 0;
}

// Plan of the fused stage-1 + q->x kernel for `batch` geometries: number of q chunks, slots per
// chunk, LDS ring.  Returns false when the shape is outside the kernel (M > 16, N > 48, no room
// for two ring blocks, workspace) -- the caller then takes the T2 path.
struct FusedPlan {
    int nchunk, qc, nbuf, ldb, wpg;
    size_t lds_bytes;
};

static int device_cu_count()
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

static bool fused_plan(int N, int M, int batch, FusedPlan* fp)
{
    if (M > 16 || N > 48 || N < 1) return false;
    // test hook: option fused_chunks = n forces n chunks of the q range (and the fused path)
    const long chunks_env = oovqe_opt(OOVQE_OPT_FUSED_CHUNKS);
    const long m2 = (long)M * M, m3 = m2 * M;
    const int n_cu = device_cu_count();
    // The persistent kernel pays off once the sweep is bandwidth-bound (>= ~6 slabs per wave on
    // every CU); below that the one-slab-per-wave T2 kernels have the shorter latency
    // (measured crossover at N = 43: 7-8 geometries).
    if (chunks_env <= 0 && (long)batch * N * N < 48L * n_cu) return false;
    // tasks (chunk, p) per geometry: about one workgroup's worth per CU when geometries are few
    long target = n_cu / batch;
    long nchunk0 = chunks_env > 0 ? chunks_env : target / N;
    if (nchunk0 < 1) nchunk0 = 1;
    // T3 [nchunk][N][M^3] + Gm [N][M^3] + Cdup [nchunk][N][N] share the 2 N^2 M^2 workspace
    const long cap = (2L * N * m2 - m3) / (m3 + N);
    if (cap < 1) return false;
    if (nchunk0 > cap) nchunk0 = cap;
    int qc = N, nchunk = 1;
    if (nchunk0 > 1) {
        qc = 4 * (int)((N + 4 * nchunk0 - 1) / (4 * nchunk0));
        nchunk = (N + qc - 1) / qc;
    }
    const int QR = (qc + 3) & ~3, CR = (nchunk - 1) * qc + QR;
    int ldb = (int)((m2 + 15) / 16 * 16);
    if (ldb % 32 == 0) ldb += 16;                  // rows lq, lq+1 of a B fragment in different bank halves
    const size_t fixed = ((size_t)CR * 16 + 64 + 8) * sizeof(double);   // C copy, sink, 5 counters per block
    const size_t block = (size_t)QR * ldb * sizeof(double);
    const size_t task_stage = (size_t)m3 * sizeof(double);            // T3 rows of one task
    const long Tg = (long)nchunk * N;
    long wpg = target < 1 ? 1 : target;
    if (wpg > Tg) wpg = Tg;
    // ring of 2-3 blocks + the T3 stage of the workgroup's ceil(Tg/wpg) tasks must fit in LDS;
    // more workgroups per geometry (fewer tasks each) if they do not
    const size_t lds_cap = 160 * 1024;
    if (fixed + 2 * block + task_stage > lds_cap) return false;
    const long max_tasks = (long)((lds_cap - fixed - 2 * block) / task_stage);
    if ((Tg + wpg - 1) / wpg > max_tasks) wpg = (Tg + max_tasks - 1) / max_tasks;
    const long ntask = (Tg + wpg - 1) / wpg;
    long nbuf = (long)((lds_cap - fixed - (size_t)ntask * task_stage) / block);
    if (nbuf > 3) nbuf = 3;
    fp->nchunk = nchunk;
    fp->qc = qc;
    fp->nbuf = (int)nbuf;
    fp->ldb = ldb;
    fp->wpg = (int)wpg;
    fp->lds_bytes = fixed + (size_t)nbuf * block + (size_t)ntask * task_stage;
    return true;
}

static int half_transform_fused_batched(const double* g_ao, const double* C, int N, int M, double* T3,
                                        double* Cdup, const FusedPlan& fp, int batch, hipStream_t st)
{
    const int ksteps = (N + 3) / 4, nrb = (N + 15) / 16;
    const int kch = ksteps <= 4 ? 4 : ksteps <= 8 ? 8 : ksteps <= 11 ? 11 : 12;
#define OOVQE_LAUNCH_FUSED(KC_, NS_)                                                              \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)half_transform_fused_kernel<KC_, NS_>, \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                               160 * 1024);                                       \
            if (e != hipSuccess) {                                                                \
                oovqe_set_error("cas_eval: hipFuncSetAttribute: %s", hipGetErrorString(e));       \
                return OOVQE_ERR_HIP;                                                             \
            }                                                                                     \
            attr_done = true;                                                                     \
        }                                                                                         \
        oovqe_note_stage1("half_transform_fused_kernel<%d,%d>", KC_, NS_);                        \
        hipLaunchKernelGGL((half_transform_fused_kernel<KC_, NS_>), dim3(fp.wpg, batch),          \
                           dim3(HALF_WAVES * 64), fp.lds_bytes, st, g_ao, C, T3, Cdup, N, M,      \
                           fp.nchunk, fp.qc, fp.nbuf, fp.ldb);                                    \
    } while (0)
    oovqe_profile_mark_start(st);
    if (kch == 4 && nrb == 1) OOVQE_LAUNCH_FUSED(4, 1);
    else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_FUSED(8, 2);
    else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_FUSED(11, 3);
    else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_FUSED(12, 3);
    else {
        oovqe_set_error("cas_eval: no fused half-transform variant for N=%d", N);
        return OOVQE_ERR_ARG;
    }
    oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_FUSED
    OOVQE_CHECK_LAUNCH("cas_eval/half_transform_fused");
    return 0;
}

static unsigned eri_slab_packed_elems(int N) { return eri_slab_pitch(N); }

extern "C" int64_t oovqe_eri_packed_size(int N)
{
    // doubles per geometry.  N <= 48: the triangle format of the batched packed-triangle kernels; N > 48: the
    // tile format of half_tiles_kernel (slabs p <= q, tile triangle of each slab, 16 x 16 tiles)
    if (N < 1) return 0;
    if (N > 48) {
        const int64_t nst = (N + 15) / 16;
        return (int64_t)N * (N + 1) / 2 * (nst * (nst + 1) / 2) * 256;
    }
    return (int64_t)N * (N + 1) / 2 * eri_slab_packed_elems(N);
}

extern "C" int oovqe_eri_pack(const double* g_ao, int N, int batch, double* packed, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(g_ao && packed, "eri_pack: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 4096 && batch >= 1 && batch <= 65535, "eri_pack: N=%d batch=%d", N, batch);
    if (N > 48) {
        hipLaunchKernelGGL(eri_tiles_pack_kernel, dim3((unsigned)((long)N * (N + 1) / 2), batch), dim3(256), 0,
                           (hipStream_t)stream, g_ao, packed, N, (N + 15) / 16);
        OOVQE_CHECK_LAUNCH("eri_pack/tiles");
        return 0;
    }
    hipLaunchKernelGGL(eri_pack_kernel, dim3((unsigned)(N * (N + 1) / 2), batch), dim3(256), 0,
                       (hipStream_t)stream, g_ao, packed, N, eri_slab_packed_elems(N));
    OOVQE_CHECK_LAUNCH("eri_pack");
    return 0;
}

// Packed-triangle stage 1 (p <-> q symmetric integrals, M <= 16, N <= 48): J [G][N(N+1)/2][M^2].
static int half_tri_batched(const double* g_ao, const double* C, int N, int M, double* J, int batch,
                            hipStream_t st, int tiled = 0, bool packed_src = false)
{
    // packed_src: g_ao is the copy made by oovqe_eri_pack (tiled == 2 only)
    OOVQE_REQUIRE(!packed_src || tiled == 2, "cas_eval: packed integrals need both symmetry flags");
    OOVQE_REQUIRE(M >= 1 && M <= 16 && N >= M && N <= 48, "cas_eval: half_tri N=%d M=%d", N, M);
    const int ksteps = (N + 3) / 4, nrb = (N + 15) / 16;
    const int kch = ksteps <= 4 ? 4 : ksteps <= 8 ? 8 : ksteps <= 11 ? 11 : 12;
    const long tri = (long)N * (N + 1) / 2;
    // Workgroups per geometry (one 8-wave workgroup is resident per CU).  When the batch divides
    // the CU count, W = n_cu / batch fills the chip in one resident round.  Otherwise a larger W
    // whose W * batch workgroups run in several rounds can waste less of the last one (96
    // geometries: W = 8, three rounds of 256, instead of W = 2 on 192 CUs); cost model: resident
    // rounds x (slab rounds per workgroup + ~3 for prologue and burst).
    const long n_cu = device_cu_count();
    const long w_max = (tri + HALF_WAVES - 1) / HALF_WAVES;
    long W = n_cu / batch;
    if (W < 1) W = 1;
    if (W > w_max) W = w_max;
    if (oovqe_opt(OOVQE_OPT_TRI_PLAIN_W) == 0) {   // (test / measurement hook: keep W = n_cu / batch)
        auto cost = [&](long w) {
            return ((long)batch * w + n_cu - 1) / n_cu * ((tri + w * HALF_WAVES - 1) / (w * HALF_WAVES) + 3);
        };
        long best = cost(W);
        for (long w = 1; w <= 16 && w <= w_max; ++w)
            if (cost(w) < best) { best = cost(w); W = w; }
    }
    const long n_rounds = (tri + W * HALF_WAVES - 1) / (W * HALF_WAVES);
    const size_t round_bytes = (size_t)HALF_WAVES * M * M * sizeof(double);
    const size_t fixed_bytes = (64 + 128) * sizeof(double);   // dump + column table
    // packed copy: three realisations of the same stage (option tri_mode; 0 = the default below):
    //   1 half_tri_kernel<.,.,2>   operand-shaped loads, two slabs per wave in registers
    //   2 half_tri_dma_kernel      LDS-DMA ring, two slots per wave
    //   3 / 4 half_tri_reg_kernel  contiguous register loads, R = 3 / 4 slabs per wave in flight
    const int tri_mode = oovqe_opt(OOVQE_OPT_TRI_MODE) ? oovqe_opt(OOVQE_OPT_TRI_MODE) : OOVQE_TRI_MODE_DEFAULT;
    const size_t ring_bytes = (size_t)HALF_WAVES * 2 * (eri_slab_packed_elems(N) + 2) * sizeof(double);
    if (packed_src && tiled == 2 && tri_mode == 2 && fixed_bytes + ring_bytes + round_bytes <= 160 * 1024) {
        long ph = (long)((160 * 1024 - fixed_bytes - ring_bytes) / round_bytes);
        if (ph > n_rounds) ph = n_rounds;
        const size_t lds_dma = fixed_bytes + ring_bytes + (size_t)ph * round_bytes;
#define OOVQE_LAUNCH_TRI_DMA(KC_, NS_)                                                            \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)half_tri_dma_kernel<KC_, NS_>,       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                                160 * 1024), "cas_eval/half_tri_dma");            \
            attr_done = true;                                                                     \
        }                                                                                         \
        oovqe_note_stage1("half_tri_dma_kernel<%d,%d>", KC_, NS_);                                \
        hipLaunchKernelGGL((half_tri_dma_kernel<KC_, NS_>), dim3((unsigned)W, batch),             \
                           dim3(HALF_WAVES * 64), lds_dma, st, g_ao, C, J, N, M, (int)ph);        \
    } while (0)
        oovqe_profile_mark_start(st);
        if (kch == 4 && nrb == 1) OOVQE_LAUNCH_TRI_DMA(4, 1);
        else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_TRI_DMA(8, 2);
        else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_TRI_DMA(11, 3);
        else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_TRI_DMA(12, 3);
        else {
            oovqe_set_error("cas_eval: no half_tri_dma variant for N=%d", N);
            return OOVQE_ERR_ARG;
        }
        oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_TRI_DMA
        OOVQE_CHECK_LAUNCH("cas_eval/half_tri_dma");
        return 0;
    }
    if (packed_src && tiled == 2 && (tri_mode == 3 || tri_mode == 4)) {
        // pieces of 1 KB per slab for the largest N of each variant: N <= 16 / 32 / 44 / 48
        const int npc = kch == 4 ? 2 : kch == 8 ? 5 : kch == 11 ? 8 : 10;
        OOVQE_REQUIRE((size_t)npc * 1024 >= eri_slab_packed_elems(N) * sizeof(double),
                      "cas_eval: half_tri_reg slab of N=%d exceeds %d KB", N, npc);
        const int R = tri_mode;
        const size_t slot_bytes = (size_t)HALF_WAVES * npc * 1024;
        long ph = (long)((160 * 1024 - fixed_bytes - slot_bytes) / round_bytes) / R * R;
        OOVQE_REQUIRE(ph >= R, "cas_eval: half_tri_reg staging does not fit LDS (M=%d)", M);
        const long nr_up = (n_rounds + R - 1) / R * R;
        if (ph > nr_up) ph = nr_up;
        const size_t lds_reg = fixed_bytes + slot_bytes + (size_t)ph * round_bytes;
#define OOVQE_LAUNCH_TRI_REG(KC_, NS_, NPC_, R_)                                                  \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)half_tri_reg_kernel<KC_, NS_, NPC_, R_>, \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                                160 * 1024), "cas_eval/half_tri_reg");            \
            attr_done = true;                                                                     \
        }                                                                                         \
        oovqe_note_stage1("half_tri_reg_kernel<%d,%d,%d,%d>", KC_, NS_, NPC_, R_);                \
        hipLaunchKernelGGL((half_tri_reg_kernel<KC_, NS_, NPC_, R_>), dim3((unsigned)W, batch),   \
                           dim3(HALF_WAVES * 64), lds_reg, st, g_ao, C, J, N, M, (int)ph);        \
    } while (0)
#define OOVQE_LAUNCH_TRI_REG_R(KC_, NS_, NPC_)                                                    \
    do {                                                                                          \
        if (R == 3) OOVQE_LAUNCH_TRI_REG(KC_, NS_, NPC_, 3);                                      \
        else OOVQE_LAUNCH_TRI_REG(KC_, NS_, NPC_, 4);                                             \
    } while (0)
        if (int rc_s1 = oovqe_stage1_enter(st)) return rc_s1;
        oovqe_profile_mark_start(st);
        if (kch == 4 && nrb == 1) OOVQE_LAUNCH_TRI_REG_R(4, 1, 2);
        else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_TRI_REG_R(8, 2, 5);
        else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_TRI_REG_R(11, 3, 8);
        else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_TRI_REG_R(12, 3, 10);
        else {
            oovqe_set_error("cas_eval: no half_tri_reg variant for N=%d", N);
            return OOVQE_ERR_ARG;
        }
        oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_TRI_REG_R
#undef OOVQE_LAUNCH_TRI_REG
        OOVQE_CHECK_LAUNCH("cas_eval/half_tri_reg");
        return oovqe_stage1_leave(st);
    }
    long phase = (long)((160 * 1024 - fixed_bytes) / round_bytes) & ~1L;   // even
    OOVQE_REQUIRE(phase >= 2, "cas_eval: half_tri staging does not fit LDS (M=%d)", M);
    if (phase > ((n_rounds + 1) & ~1L)) phase = (n_rounds + 1) & ~1L;
    const size_t lds_bytes = fixed_bytes + (size_t)phase * round_bytes;
#define OOVQE_LAUNCH_TRI2(KC_, NS_, RS_)                                                          \
    do {                                                                                          \
        static bool attr_done = false;                                                            \
        if (!attr_done) {                                                                         \
            OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)half_tri_kernel<KC_, NS_, RS_>,      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                                160 * 1024), "cas_eval/half_tri");                \
            attr_done = true;                                                                     \
        }                                                                                         \
        oovqe_note_stage1("half_tri_kernel<%d,%d,%d>", KC_, NS_, RS_);                            \
        hipLaunchKernelGGL((half_tri_kernel<KC_, NS_, RS_>), dim3((unsigned)W, batch),            \
                           dim3(HALF_WAVES * 64), lds_bytes, st, g_ao, C, J, N, M, (int)phase,    \
                           tiled);                                                                \
    } while (0)
#define OOVQE_LAUNCH_TRI(KC_, NS_)                                                                \
    do {                                                                                          \
        if (tiled == 2 && packed_src) OOVQE_LAUNCH_TRI2(KC_, NS_, 2);                             \
        else if (tiled == 2) OOVQE_LAUNCH_TRI2(KC_, NS_, 1);                                      \
        else OOVQE_LAUNCH_TRI2(KC_, NS_, 0);                                                      \
    } while (0)
    oovqe_profile_mark_start(st);
    if (kch == 4 && nrb == 1) OOVQE_LAUNCH_TRI(4, 1);
    else if (kch == 8 && nrb == 2) OOVQE_LAUNCH_TRI(8, 2);
    else if (kch == 11 && nrb == 3) OOVQE_LAUNCH_TRI(11, 3);
    else if (kch == 12 && nrb == 3) OOVQE_LAUNCH_TRI(12, 3);
    else {
        oovqe_set_error("cas_eval: no half_tri variant for N=%d", N);
        return OOVQE_ERR_ARG;
    }
    oovqe_profile_mark_stop(st);
#undef OOVQE_LAUNCH_TRI
#undef OOVQE_LAUNCH_TRI2
    OOVQE_CHECK_LAUNCH("cas_eval/half_tri");
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

static int cas_energy_gradient_impl(const double* Gm, const double* hmo, const double* gamma,
                                    const double* Gamma, int nrdm, double nuc, const double* nuc_dev, int N,
                                    int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                    int n_kappa, double* c0, double* c1, double* c2, double* E,
                                    double* fock, double* gmat, double* gvec, double* dE,
                                    oovqe_stream_t stream);

extern "C" int oovqe_cas_energy_gradient(const double* Gm, const double* hmo, const double* gamma,
                                         const double* Gamma, int nrdm, double nuc, int N, int n_occ,
                                         int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                         int n_kappa, double* c0, double* c1, double* c2, double* E,
                                         double* fock, double* gmat, double* gvec, double* dE,
                                         oovqe_stream_t stream)
{
    return cas_energy_gradient_impl(Gm, hmo, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row,
                                    kap_col, n_kappa, c0, c1, c2, E, fock, gmat, gvec, dE, stream);
}

static int cas_energy_gradient_impl(const double* Gm, const double* hmo, const double* gamma,
                                    const double* Gamma, int nrdm, double nuc, const double* nuc_dev, int N,
                                    int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
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
                       c2, E, fock, gmat, gvec, dE, nuc_dev);
    OOVQE_CHECK_LAUNCH("cas_energy_gradient");
    return 0;
}

// The same stage from MO integrals in memory with a workgroup per (n, RDM set) and the final
// assembly kernel; Fcol [nrdm][M][N], Epart [nrdm][N], Cpart [N] are scratch.
static int cas_energy_gradient_rows(const double* Gm, const double* hmo, const double* gamma,
                                    const double* Gamma, int nrdm, double nuc, const double* nuc_dev, int N,
                                    int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                    int n_kappa, double* Fcol, double* Epart, double* Cpart, double* c0,
                                    double* c1, double* c2, double* E, double* fock, double* gmat,
                                    double* gvec, double* dE, oovqe_stream_t stream)
{
    const int M = n_occ + ncas;
    OOVQE_REQUIRE(M <= 64 && nrdm <= 65535, "cas_energy_gradient: n_occ + ncas = %d, nrdm = %d", M, nrdm);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = fock_rows_lds_elems(n_occ, ncas) * sizeof(double);
    OOVQE_REQUIRE(lds_bytes <= FROW_LDS_MAX, "cas_energy_gradient: %zu B of LDS", lds_bytes);
    static bool attr_done = false;
    if (!attr_done) {
        OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)fock_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)FROW_LDS_MAX), "cas_energy_gradient");
        attr_done = true;
    }
    hipLaunchKernelGGL(fock_rows_kernel, dim3(N, nrdm), dim3(FROW_THREADS), lds_bytes, st, Gm, hmo, gamma, Gamma,
                       N, n_occ, ncas, Fcol, Epart, Cpart, c1, c2);
    OOVQE_CHECK_LAUNCH("cas_energy_gradient/rows");
    hipLaunchKernelGGL(cas_final_kernel, dim3(1), dim3(512), 0, st, Fcol, Epart, Cpart, nuc, nrdm, N, M,
                       kap_row, kap_col, n_kappa, c0, E, gvec, dE, fock, gmat, nuc_dev, (size_t)0);
    OOVQE_CHECK_LAUNCH("cas_energy_gradient/final");
    return 0;
}

extern "C" int64_t oovqe_cas_energy_gradient_work_size(int N, int n_occ, int ncas, int nrdm)
{
    const int64_t M = n_occ + ncas;
    return (int64_t)nrdm * (M + 1) * N + N;      // Fcol [nrdm][M][N] | Epart [nrdm][N] | Cpart [N]
}

extern "C" int oovqe_cas_energy_gradient_ws(const double* Gm, const double* hmo, const double* gamma,
                                            const double* Gamma, int nrdm, double nuc, int N, int n_occ,
                                            int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                            int n_kappa, double* c0, double* c1, double* c2, double* E,
                                            double* fock, double* gmat, double* gvec, double* dE,
                                            double* work, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(Gm && hmo && gamma && Gamma && c0 && c1 && c2 && E && gvec && work,
                  "cas_energy_gradient: null pointer");
    OOVQE_REQUIRE(nrdm >= 1 && N >= 1 && n_occ >= 0 && ncas >= 1 && n_occ + ncas <= N,
                  "cas_energy_gradient: bad sizes");
    OOVQE_REQUIRE(n_kappa == 0 || (kap_row && kap_col), "cas_energy_gradient: null index table");
    OOVQE_REQUIRE(nrdm == 1 || dE, "cas_energy_gradient: dE required when nrdm > 1");
    const size_t M = (size_t)n_occ + ncas;
    if (M > 64 || fock_rows_lds_elems(n_occ, ncas) * sizeof(double) > FROW_LDS_MAX)   // one-workgroup kernel
        return cas_energy_gradient_impl(Gm, hmo, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row,
                                        kap_col, n_kappa, c0, c1, c2, E, fock, gmat, gvec, dE, stream);
    double* Fc = work;
    double* Ep = Fc + (size_t)nrdm * M * N;
    double* Cp = Ep + (size_t)nrdm * N;
    return cas_energy_gradient_rows(Gm, hmo, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row, kap_col,
                                    n_kappa, Fc, Ep, Cp, c0, c1, c2, E, fock, gmat, gvec, dE, stream);
}

// LDS of cas_column_kernel without the RDM sets (U[n] and g_mo[n] resident): decides between the
// column kernel and the staged kernels for large N * M^2
static size_t column_base_bytes(int N, int M)
{
    const size_t m2 = (size_t)M * M, m3 = m2 * M;
    return ((size_t)N * m2 + (size_t)N * M + m3 + N + M + M + N + (size_t)4 * N) * sizeof(double);
}

static bool column_fits(int N, int M, int ncas)
{
    const size_t na2 = (size_t)ncas * ncas;
    return column_base_bytes(N, M) + (na2 + na2 * na2) * sizeof(double) <= 160 * 1024;
}

// Batched CAS path: `batch` geometries of identical shape, every per-geometry array stacked.
// Outputs c0/c1/c2/E/gvec/dE of geometry g live at pointer + g * out_stride (doubles).
// Where the circuit launch of an evaluation may leave W = C^T h_ao [G][N][N] for the panel kernel: the T3 block of
// the packed-triangle path's workspace, which its one-launch q -> x / p -> n kernel does not use.  False when the
// call will not take that path (the decision tree of cas_eval_batched below).
extern "C" int64_t oovqe_oo_eval_out_size(int n_theta, int n_kappa, int ncas, int derivatives);

static bool cas_w_block(int N, int M, int batch, unsigned eri_flags, double* work, double** W)
{
    if (N > 48 || oovqe_opt(OOVQE_OPT_CAS_UNFUSED) != 0 || oovqe_opt(OOVQE_OPT_SYM_MIRROR) != 0 ||
        oovqe_opt(OOVQE_OPT_SYM_SIMPLE) != 0 || oovqe_opt(OOVQE_OPT_SYM_TWO_STEP) != 0 ||
        oovqe_opt(OOVQE_OPT_PANEL_NO_W) != 0 || (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) == 0)
        return false;
    FusedPlan fp;
    if (!fused_plan(N, M, batch, &fp)) return false;
    const long m2 = (long)M * M, m3 = m2 * M;
    const long tri = (long)N * (N + 1) / 2;
    const long nty16 = (m2 + 15) / 16 * 16;
    if (tri * nty16 + 2 * (long)N * m3 > 2 * (long)N * N * m2) return false;
    *W = work + (size_t)batch * tri * nty16;
    return true;
}

static int cas_eval_batched(const double* g_ao, const double* h_ao, const double* C,
                            const double* gamma, const double* Gamma, int nrdm, double nuc,
                            const double* nuc_arr, int N, int n_occ, int ncas, const int32_t* kap_row,
                            const int32_t* kap_col, int n_kappa, double* work, double* c0, double* c1,
                            double* c2, double* E, double* gvec, double* dE, double* fock,
                            double* gmat, double* Gm, double* hmo, int batch, size_t out_stride,
                            oovqe_stream_t stream, const oovqe_circuit_job_t* cj = nullptr,
                            unsigned eri_flags = 0, const double* g_packed = nullptr,
                            const double* T2_ready = nullptr, bool w_ready = false, hipEvent_t rdm_event = nullptr)
{
    // rdm_event: recorded on the stream where gamma / Gamma are complete when the circuit rides along (cj): behind the
    // q -> x / p -> n launch -- another stream of the caller (the orbital Hessian's assembly) waits for the RDMs, not
    // for the end of this call
    // w_ready: the caller's circuit launch has left W = C^T h_ao [G][N][N] in this workspace's T3 block
    // (cas_w_block: packed-triangle path, one-launch q -> x / p -> n kernel)
    // T2_ready [G][N][N][M][M]: the caller has stage 1's result in memory (the Hessian call); the
    // packed-triangle path then builds its J from it instead of reading the integrals again
    // cj: circuit + RDM evaluations that produce gamma / Gamma; they ride along the p -> n
    // contraction launch (the caller has checked oovqe_contract_hosts_circuit for this shape)
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
    // fused path: T3 = stage 1 + q -> x in one kernel, then the small p -> n contraction gives
    // g_mo[n,x,y,z] directly; T2 path (M > 16 or N > 48): T2, U = C^T T2, q -> x in the column kernel
    const bool unfused_env = oovqe_opt(OOVQE_OPT_CAS_UNFUSED) != 0;   // test hook, read per call
    FusedPlan fp;
    const bool fused = !unfused_env && fused_plan(N, M, batch, &fp);
    const double* Gm_in = nullptr;
    const double* Wpre = nullptr;
    // p <-> q symmetric integrals (verified by the caller): only the slabs p <= q are read
    const bool pq_sym = (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) != 0;
    // r <-> s symmetric as well: J[p,q,y,z] == J[p,q,z,y], the packed path keeps the columns y <= z
    const bool rs_sym = (eri_flags & OOVQE_ERI_RS_SYMMETRIC) != 0 && oovqe_opt(OOVQE_OPT_SYM_NO_RS) == 0;
    const int half_sym = pq_sym ? SYM_MIRROR : SYM_FULL;
    const long tri = (long)N * (N + 1) / 2;
    const long nty16 = (m2 + 15) / 16 * 16;               // tile-major J rows are padded to 16
    const bool sym_packed = fused && pq_sym && oovqe_opt(OOVQE_OPT_SYM_MIRROR) == 0 &&
                            tri * nty16 + 2 * (long)N * m3 <= 2 * (long)N * N * m2;
    if (sym_packed) {
        // packed triangle J (instead of the fused kernel: its q -> x contraction needs whole rows
        // of slabs), then the small q -> x kernel; from T3 on the same launches as the fused path
        double* Jp = work;                                       // [G][tri][M^2] or [G][nty][tri][16]
        double* T3 = Jp + nb * tri * nty16;                      // [G][N][M^3]
        double* Gmw = T3 + nb * N * m3;                          // [G][N][M^3]
        // test hooks: OOVQE_SYM_SIMPLE = the one-slab-per-wave kernel (direct stores) instead of the
        // persistent one; OOVQE_SYM_TWO_STEP = q -> x kernel, then K1, instead of the one-launch
        // kernel (both on the row-major J)
        const bool simple = oovqe_opt(OOVQE_OPT_SYM_SIMPLE) != 0;
        const bool two_step = simple || oovqe_opt(OOVQE_OPT_SYM_TWO_STEP) != 0;
        if (simple) {
            if ((rc = half_transform_batched(g_ao, C, N, M, Jp, batch, stream, SYM_PACKED))) return rc;
        } else {
            // the packed copy of the integrals (oovqe_eri_pack) is streamed when the caller holds one
            const bool use_pk = !two_step && rs_sym && g_packed != nullptr;
            if (T2_ready && !two_step) {
                const unsigned nbk = (unsigned)((tri + 15) / 16);       // 16 rows of the triangle per workgroup
                oovqe_profile_mark_start_l(st, 0);
                hipLaunchKernelGGL(j_from_t2_kernel, dim3(nbk, batch), dim3(256), 0, st, T2_ready, Jp, N, M,
                                   rs_sym ? 1 : 0);
                oovqe_profile_mark_stop(st);
                OOVQE_CHECK_LAUNCH("cas_eval/j_from_t2");
            } else if ((rc = half_tri_batched(use_pk ? g_packed : g_ao, C, N, M, Jp, batch, st,
                                              two_step ? 0 : rs_sym ? 2 : 1, use_pk)))
                return rc;
        }
        if (!two_step) {
            if ((rc = sym_gm_batched(Jp, C, Gmw, N, M, batch, st, cj, rs_sym))) return rc;
            if (rdm_event && cj) {
                OOVQE_CHECK_HIP(hipEventRecord(rdm_event, st), "cas_eval: hipEventRecord");
                rdm_event = nullptr;
            }
            if (w_ready) {                                       // (the block this path leaves unused)
                double* wchk = nullptr;
                OOVQE_REQUIRE(cas_w_block(N, M, batch, eri_flags, work, &wchk) && wchk == T3,
                              "cas_eval: W = C^T h was promised for another path");
                Wpre = T3;
            }
        } else {
            if ((rc = sym_q_contract_batched(Jp, C, T3, N, M, batch, st))) return rc;
            oovqe_profile_mark_start_l(st, 2);
            if ((rc = oovqe_mode_contract_batched_circ(T3, C, Gmw, 1, N, N, m3, N, 0, batch, (long)N * m3,
                                                       (long)N * N, (long)N * m3, st, cj)))
                return rc;
            oovqe_profile_mark_stop(st);
        }
        Gm_in = Gmw;
    } else if (fused) {
        double* T3 = work;                                       // [G][nchunk][N][M^3]
        double* Gmw = T3 + nb * fp.nchunk * N * m3;              // [G][N][M^3]
        double* Cdup = Gmw + nb * N * m3;                        // [G][nchunk][N][N]  (nchunk > 1)
        if ((rc = half_transform_fused_batched(g_ao, C, N, M, T3, Cdup, fp, batch, st))) return rc;
        // Gm[n,(x y z)] = sum_{c,p} C[p,n] T3[c,p,(x y z)]
        oovqe_profile_mark_start_l(st, 2);
        const long K = (long)fp.nchunk * N;
        if ((rc = oovqe_mode_contract_batched_circ(T3, fp.nchunk > 1 ? Cdup : C, Gmw, 1, (int)K, N, m3, N, 0,
                                                   batch, K * m3, fp.nchunk > 1 ? K * N : (long)N * N,
                                                   (long)N * m3, st, cj)))
            return rc;
        oovqe_profile_mark_stop(st);
        Gm_in = Gmw;
    } else if (!column_fits(N, M, ncas)) {
        // Large N * M^2 (many occupied orbitals): neither U[n] nor the panel fits LDS.  Staged path
        // per geometry: T2 (batched) -> K1: q -> x, p -> n -> g_mo[n,x,y,z] and h_mo in memory ->
        // fock_kernel (one workgroup per RDM set, streaming g_mo from L2).  Same outputs.
        OOVQE_REQUIRE(M < N, "cas_eval: the staged path needs at least one virtual orbital (N=%d M=%d)", N, M);
        if (cj) {   // the circuit workgroups cannot ride along here: own launch
            if ((rc = oovqe_circuit_rdms(cj->theta, cj->n_theta, cj->gates, cj->n_gates, cj->n_qubits,
                                         cj->ncas, cj->init_index, cj->n_tan > 0, batch, nullptr, nullptr,
                                         cj->gamma, cj->Gamma, nullptr, stream)))
                return rc;
        }
        if ((rc = half_transform_batched(g_ao, C, N, M, T2, batch, stream, half_sym, rs_sym, nullptr,
                                         N > 48 ? g_packed : nullptr)))
            return rc;
        const size_t na2s = (size_t)ncas * ncas, na4s = na2s * na2s;
        for (int g = 0; g < batch; ++g) {
            const size_t gi = (size_t)g;
            double* T2g = T2 + gi * N * N * m2;
            double* wk = U + gi * N * N * m2;                  // T3 [N][M^3] + Y [N][M]
            double* Gmg = T2g;                                 // T2 of this geometry is dead after q -> x
            double* hmog = Fcol + gi * nrdm * M * N;           // [N][M]
            if ((rc = oovqe_cas_finish_transform(T2g, h_ao + gi * N * N, C + gi * N * N, N, M, Gmg, hmog,
                                                 wk, stream)))
                return rc;
            // scratch behind T3 and Y in this geometry's U block (N^2 M^2 >= N M^3 + N M + the three below)
            double* Fc = wk + (size_t)N * m3 + (size_t)N * M;
            double* Ep = Fc + (size_t)nrdm * M * N;
            double* Cp = Ep + (size_t)nrdm * N;
            const bool rows_fit = (size_t)N * m3 + (size_t)N * M + (size_t)nrdm * (M + 1) * N + N <=
                                  (size_t)N * N * m2 && M <= 64 &&
                                  fock_rows_lds_elems(n_occ, ncas) * sizeof(double) <= FROW_LDS_MAX;
            if (rows_fit)
                rc = cas_energy_gradient_rows(
                    Gmg, hmog, gamma + gi * nrdm * na2s, Gamma + gi * nrdm * na4s, nrdm, nuc,
                    nuc_arr ? nuc_arr + g : nullptr, N, n_occ, ncas, kap_row, kap_col, n_kappa, Fc, Ep, Cp,
                    c0 + gi * out_stride, c1 + gi * out_stride, c2 + gi * out_stride, E + gi * out_stride,
                    fock ? fock + gi * N * N : nullptr, gmat ? gmat + gi * N * N : nullptr,
                    gvec + gi * out_stride, dE ? dE + gi * out_stride : nullptr, stream);
            else
                rc = cas_energy_gradient_impl(
                    Gmg, hmog, gamma + gi * nrdm * na2s, Gamma + gi * nrdm * na4s, nrdm, nuc,
                    nuc_arr ? nuc_arr + g : nullptr, N, n_occ, ncas, kap_row, kap_col, n_kappa,
                    c0 + gi * out_stride, c1 + gi * out_stride, c2 + gi * out_stride, E + gi * out_stride,
                    fock ? fock + gi * N * N : nullptr, gmat ? gmat + gi * N * N : nullptr,
                    gvec + gi * out_stride, dE ? dE + gi * out_stride : nullptr, stream);
            if (rc) return rc;
            if (Gm)
                OOVQE_CHECK_HIP(hipMemcpyAsync(Gm + gi * N * m3, Gmg, (size_t)N * m3 * sizeof(double),
                                               hipMemcpyDeviceToDevice, st), "cas_eval: copy g_mo");
            if (hmo)
                OOVQE_CHECK_HIP(hipMemcpyAsync(hmo + gi * N * M, hmog, (size_t)N * M * sizeof(double),
                                               hipMemcpyDeviceToDevice, st), "cas_eval: copy h_mo");
        }
        return 0;
    } else {
        if ((rc = half_transform_batched(g_ao, C, N, M, T2, batch, stream, half_sym, rs_sym, nullptr,
                                         N > 48 ? g_packed : nullptr)))
            return rc;
        // U[n,(q y z)] = sum_p C[p,n] T2[p,(q y z)]
        oovqe_profile_mark_start_l(st, 2);
        if ((rc = oovqe_mode_contract_batched_circ(T2, C, U, 1, N, N, (long)N * m2, N, 0, batch,
                                                   (long)N * N * m2, (long)N * N, (long)N * N * m2, st,
                                                   cj)))
            return rc;
        oovqe_profile_mark_stop(st);
    }
    const size_t na2 = (size_t)ncas * ncas;
    const size_t set_bytes = (na2 + na2 * na2) * sizeof(double);
    const size_t lds_cap = 160 * 1024;
    if (fused || sym_packed) {
        // panel kernel: npan general indices per workgroup, about one resident round of workgroups
        const size_t fixed_bytes = ((size_t)N * M + (Wpre ? 0 : (size_t)N * N)) * sizeof(double);
        const size_t per_n = ((size_t)m3 + (Wpre ? 1 : 2) * (size_t)N + 2 * (size_t)M) * sizeof(double);
        OOVQE_REQUIRE(fixed_bytes + per_n + set_bytes <= lds_cap, "cas_eval: N=%d M=%d needs %zu B of LDS",
                      N, M, fixed_bytes + per_n + set_bytes);
        long npan = ((long)N * batch + 383) / 384;
        const long npan_max = (long)((lds_cap - fixed_bytes - set_bytes) / per_n);
        if (npan > npan_max) npan = npan_max;
        if (npan > 8) npan = 8;   // (two workgroups per CU: 40 us against 47 us with panels of 16 at 256 geometries)
        if (Wpre) {
            // without h_ao in LDS a panel of this many general indices leaves room for THREE workgroups per CU
            // (N = 43, M = 9: 7 indices, 51 KB; 39 -> 31 us at 256 geometries)
            const long n3 = (long)((lds_cap / 3 - fixed_bytes - set_bytes) / per_n);
            if (n3 >= 4 && n3 < npan) npan = n3;
        }
        if (oovqe_opt(OOVQE_OPT_PANEL_ROWS) > 0) npan = oovqe_opt(OOVQE_OPT_PANEL_ROWS);   // measurement hook
        if (npan < 1) npan = 1;
        int rdm_chunk = (int)((lds_cap - fixed_bytes - (size_t)npan * per_n) / set_bytes);
        if (rdm_chunk > nrdm) rdm_chunk = nrdm;
        const size_t lds_bytes = fixed_bytes + (size_t)npan * per_n + (size_t)rdm_chunk * set_bytes;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)cas_panel_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
            if (e != hipSuccess) {
                oovqe_set_error("cas_eval: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return OOVQE_ERR_HIP;
            }
            attr_done = true;
        }
        oovqe_profile_mark_start_l(st, 3);
        const unsigned npanels = (unsigned)((N + npan - 1) / npan);
        const bool xcd_grid = batch > 1 && oovqe_opt(OOVQE_OPT_GM_PLAIN_GRID) == 0;
        hipLaunchKernelGGL(cas_panel_kernel,
                           xcd_grid ? dim3(npanels * (unsigned)((batch + 7) / 8 * 8)) : dim3(npanels, batch),
                           dim3(PAN_THREADS), lds_bytes, st, Gm_in, h_ao, C, gamma, Gamma, nrdm, N, n_occ,
                           ncas, (int)npan, Fcol, Epart, Cpart, c1, c2, Gm, hmo, out_stride, rdm_chunk,
                           xcd_grid ? batch : 0, Wpre);
        oovqe_profile_mark_stop(st);
        OOVQE_CHECK_LAUNCH("cas_eval/panel");
    } else {
        const size_t base_bytes = ((size_t)N * m2 + (size_t)N * M + m3 + N + M + M + N + (size_t)4 * N) *
                                  sizeof(double);
        OOVQE_REQUIRE(base_bytes + set_bytes <= lds_cap, "cas_eval: N=%d M=%d needs %zu B of LDS", N, M,
                      base_bytes + set_bytes);
        int rdm_chunk = (int)((lds_cap - base_bytes) / set_bytes);
        if (rdm_chunk > nrdm) rdm_chunk = nrdm;
        const size_t lds_bytes = base_bytes + (size_t)rdm_chunk * set_bytes;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)cas_column_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
            if (e != hipSuccess) {
                oovqe_set_error("cas_eval: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return OOVQE_ERR_HIP;
            }
            attr_done = true;
        }
        oovqe_profile_mark_start_l(st, 3);
        // many large RDM sets (a derivative evaluation of a big active space): their chunks are dealt to
        // zsplit workgroups per n, about one resident round of workgroups in all
        int zsplit = 1;
        if (ncas * ncas * ncas >= 128 && rdm_chunk > 0) {
            const int nchunks = (nrdm + rdm_chunk - 1) / rdm_chunk;
            zsplit = (int)((long)device_cu_count() * 2 / ((long)N * batch));
            if (zsplit > nchunks) zsplit = nchunks;
            if (zsplit > 64) zsplit = 64;
            if (zsplit < 1) zsplit = 1;
        }
        hipLaunchKernelGGL(cas_column_kernel, dim3(N, batch, zsplit), dim3(COL_THREADS), lds_bytes, st, U, h_ao, C,
                           gamma, Gamma, nrdm, N, n_occ, ncas, Fcol, Epart, Cpart, c1, c2, Gm, hmo,
                           out_stride, rdm_chunk);
        oovqe_profile_mark_stop(st);
        OOVQE_CHECK_LAUNCH("cas_eval/column");
    }
    oovqe_profile_mark_start_l(st, 4);
    hipLaunchKernelGGL(cas_final_kernel, dim3(batch), dim3(512), 0, st, Fcol, Epart, Cpart, nuc, nrdm,
                       N, M, kap_row, kap_col, n_kappa, c0, E, gvec, dE, fock, gmat, nuc_arr,
                       out_stride);
    oovqe_profile_mark_stop(st);
    OOVQE_CHECK_LAUNCH("cas_eval/final");
    if (rdm_event) OOVQE_CHECK_HIP(hipEventRecord(rdm_event, st), "cas_eval: hipEventRecord");   // (no earlier point on this path)
    return 0;
}

extern "C" int oovqe_cas_eval(const double* g_ao, const double* h_ao, const double* C,
                              const double* gamma, const double* Gamma, int nrdm, double nuc, int N,
                              int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                              int n_kappa, double* work, double* c0, double* c1, double* c2,
                              double* E, double* gvec, double* dE, double* fock, double* gmat,
                              double* Gm, double* hmo, unsigned eri_flags, oovqe_stream_t stream)
{
    return cas_eval_batched(g_ao, h_ao, C, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row,
                            kap_col, n_kappa, work, c0, c1, c2, E, gvec, dE, fock, gmat, Gm, hmo, 1, 0,
                            stream, nullptr, eri_flags);
}

extern "C" int oovqe_cas_eval_packed(const double* g_ao, const double* h_ao, const double* C,
                                     const double* gamma, const double* Gamma, int nrdm, double nuc, int N,
                                     int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                     int n_kappa, double* work, double* c0, double* c1, double* c2,
                                     double* E, double* gvec, double* dE, double* fock, double* gmat,
                                     double* Gm, double* hmo, unsigned eri_flags, const double* g_packed,
                                     oovqe_stream_t stream)
{
    OOVQE_REQUIRE(!g_packed || eri_flags == (OOVQE_ERI_PQ_SYMMETRIC | OOVQE_ERI_RS_SYMMETRIC),
                  "cas_eval: a packed copy needs both symmetry flags");
    return cas_eval_batched(g_ao, h_ao, C, gamma, Gamma, nrdm, nuc, nullptr, N, n_occ, ncas, kap_row,
                            kap_col, n_kappa, work, c0, c1, c2, E, gvec, dE, fock, gmat, Gm, hmo, 1, 0,
                            stream, nullptr, eri_flags, g_packed);
}

// The CAS path for a STACK of geometries from given RDM sets (the circuit lives elsewhere: the sector engine of large
// registers): geometry g reads gamma [g][nrdm][a^2], Gamma [g][nrdm][a^4] and writes the packed outputs of
// oovqe_oo_eval_batch for n_theta = nrdm - 1 at out + g * oovqe_oo_eval_out_size(nrdm - 1, n_kappa, ncas, nrdm > 1).
extern "C" int oovqe_cas_eval_batch(const double* g_ao, const double* h_ao, const double* C, const double* gamma,
                                    const double* Gamma, int nrdm, const double* nuc, int N, int n_occ, int ncas,
                                    const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int batch,
                                    double* work, double* out, double* fock, unsigned eri_flags,
                                    const double* g_packed, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(nuc && out, "cas_eval_batch: null pointer");
    OOVQE_REQUIRE(nrdm >= 1 && batch >= 1, "cas_eval_batch: nrdm = %d, batch = %d", nrdm, batch);
    OOVQE_REQUIRE(!g_packed || eri_flags == (OOVQE_ERI_PQ_SYMMETRIC | OOVQE_ERI_RS_SYMMETRIC),
                  "cas_eval_batch: a packed copy needs both symmetry flags");
    const size_t na2 = (size_t)ncas * ncas;
    const size_t out_stride = (size_t)oovqe_oo_eval_out_size(nrdm - 1, n_kappa, ncas, nrdm > 1);
    const int n_t = nrdm > 1 ? nrdm - 1 : 1;
    double* c0 = out;
    double* E = out + 1;
    double* dE = out + 2;
    double* gvec = dE + n_t;
    double* c1 = gvec + (size_t)nrdm * n_kappa;
    double* c2 = c1 + na2;
    return cas_eval_batched(g_ao, h_ao, C, gamma, Gamma, nrdm, 0.0, nuc, N, n_occ, ncas, kap_row, kap_col, n_kappa,
                            work, c0, c1, c2, E, gvec, dE, fock, nullptr, nullptr, nullptr, batch, out_stride, stream,
                            nullptr, eri_flags, g_packed);
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
                           unsigned eri_flags, oovqe_stream_t stream, const double* g_packed = nullptr,
                           double* fock = nullptr, const double* T2_ready = nullptr, hipEvent_t rdm_event = nullptr)
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
    // The circuit + RDM step is independent of the integral transform until the Fock stage.  When
    // it is the one-workgroup kind and the p -> n contraction of this shape is a single-chunk K1
    // launch, its workgroups ride along that launch (one launch and ~12 us of latency less per
    // evaluation); otherwise it is launched here.  OOVQE_NO_RIDE=1 forces the separate launch.
    oovqe_circuit_job_t cj;
    bool ride = false;
    if (oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates) && oovqe_opt(OOVQE_OPT_NO_RIDE) != 1) {
        const int M = n_occ + ncas;
        const long m2 = (long)M * M, m3 = m2 * M;
        cj.theta = theta;
        cj.gates = gates;
        cj.gamma = gamma;
        cj.Gamma = Gamma;
        cj.n_theta = n_theta;
        cj.n_gates = n_gates;
        cj.n_qubits = n_qubits;
        cj.ncas = ncas;
        cj.n_tan = nvec - 1;
        cj.count = batch;
        cj.init_index = init_index;
        cj.lds_bytes = oovqe_small_circuit_lds_bytes(n_qubits, ncas, nvec, n_gates);
        FusedPlan fp;
        const bool fused = oovqe_opt(OOVQE_OPT_CAS_UNFUSED) == 0 && fused_plan(N, M, batch, &fp);
        const long K = fused ? (long)fp.nchunk * N : N, B = fused ? m3 : (long)N * m2;
        ride = cj.lds_bytes <= 64 * 1024 && K <= 0x7fffffffL &&
               oovqe_contract_hosts_circuit(1, (int)K, N, B, 0, batch);
        // On the packed-triangle path the host launch is sym_gm_kernel, two workgroups per CU: the circuit
        // workgroups are its longest (a latency chain through the gates and the RDM products, slowed further by the
        // matrix-core workgroups they share a CU with), and once the grid is beyond 1.5 resident rounds they end
        // the launch alone.  Measured per call (tools/ride_probe.py, riding / own launch): 192 geometries 363.9 /
        // 370.2 us, 224: 427.8 / 420.6, 256: 471.4 / 464.5, 384: 734.6 / 705.0, 512: 983.4 / 946.9.
        if (ride && fused && (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) != 0 && oovqe_opt(OOVQE_OPT_SYM_MIRROR) == 0 &&
            oovqe_opt(OOVQE_OPT_NO_RIDE) != 2) {
            const bool rs = (eri_flags & OOVQE_ERI_RS_SYMMETRIC) != 0 && oovqe_opt(OOVQE_OPT_SYM_NO_RS) == 0;
            const long ntile = ((rs ? (long)M * (M + 1) / 2 : m2) + 15) / 16;
            if ((ntile + 1) * batch > 3L * device_cu_count()) ride = false;
        }
    }
    bool w_ready = false;
    if (!ride) {
        // the circuit as a launch of its own: W = C^T h_ao of every geometry comes from extra workgroups of that
        // launch when the panel kernel will take it (packed-triangle path, small circuit, T2 not from a caller)
        double* Wpre = nullptr;
        if (oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates) && batch <= 32767)
            w_ready = cas_w_block(N, n_occ + ncas, batch, eri_flags, cas_work, &Wpre);
        oovqe_profile_mark_start_l((hipStream_t)stream, 1);
        int rc = oovqe_circuit_rdms_w(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index,
                                      derivatives, batch, psi, derivatives ? dpsi : nullptr, gamma, Gamma,
                                      rwork, w_ready ? h_ao : nullptr, w_ready ? C : nullptr, N,
                                      w_ready ? Wpre : nullptr, stream);
        if (rc) return rc;
        oovqe_profile_mark_stop((hipStream_t)stream);
        if (rdm_event) {                                  // the RDMs are complete behind the circuit's own launch
            OOVQE_CHECK_HIP(hipEventRecord(rdm_event, (hipStream_t)stream), "oo_eval: hipEventRecord");
            rdm_event = nullptr;
        }
    }
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
                            kap_col, n_kappa, cas_work, c0, c1, c2, E, gvec, dE, fock, nullptr,
                            nullptr, nullptr, batch, out_stride, stream, ride ? &cj : nullptr, eri_flags,
                            g_packed, T2_ready, w_ready, rdm_event);
}

// hessian.hip (oovqe_oo_hessian_batch): the batched evaluation with the generalized Fock matrices
// [G][N][N] as an extra output; the RDM sets stay at the head of `work` (gamma [G][nvec][a^2], then
// Gamma [G][nvec][a^4]) for the orbital-Hessian stage that follows.
int oovqe_oo_eval_batched_impl(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                               int n_qubits, uint32_t init_index, const double* g_ao, const double* h_ao,
                               const double* C, const double* nuc_arr, int N, int n_occ, int ncas,
                               const int32_t* kap_row, const int32_t* kap_col, int n_kappa, int derivatives,
                               int batch, double* work, double* out, unsigned eri_flags,
                               oovqe_stream_t stream, const double* g_packed, double* fock,
                               const double* T2_ready, hipEvent_t rdm_event)
{
    return oo_eval_batched(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, 0.0, nuc_arr,
                           N, n_occ, ncas, kap_row, kap_col, n_kappa, derivatives, batch, work, out,
                           eri_flags, stream, g_packed, fock, T2_ready, rdm_event);
}

// hessian.hip: stage 1 (T2[p,q,y,z]) for a stack of geometries, reading only the slabs p <= q when the
// caller vouches for the p<->q symmetry
int oovqe_half_transform_batched_impl(const double* g_ao, const double* C, int N, int M, double* T2,
                                      int batch, unsigned eri_flags, oovqe_stream_t stream, double* Vk_tri)
{
    // Vk_tri [G][N(N+1)/2][N][M] (optional; p <-> q symmetric integrals, N <= 48): see half_transform_kernel
    const bool pq = (eri_flags & OOVQE_ERI_PQ_SYMMETRIC) != 0;
    const bool rs = (eri_flags & OOVQE_ERI_RS_SYMMETRIC) != 0 && oovqe_opt(OOVQE_OPT_SYM_NO_RS) == 0;
    return half_transform_batched(g_ao, C, N, M, T2, batch, stream, pq ? SYM_MIRROR : SYM_FULL, rs, Vk_tri);
}

extern "C" int oovqe_oo_eval(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates,
                             int n_qubits, uint32_t init_index, const double* g_ao,
                             const double* h_ao, const double* C, double nuc, int N, int n_occ,
                             int ncas, const int32_t* kap_row, const int32_t* kap_col, int n_kappa,
                             int derivatives, double* work, double* out, unsigned eri_flags,
                             oovqe_stream_t stream)
{
    return oo_eval_batched(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, nuc,
                           nullptr, N, n_occ, ncas, kap_row, kap_col, n_kappa, derivatives, 1, work,
                           out, eri_flags, stream);
}

extern "C" int oovqe_oo_eval_batch(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                   int n_gates, int n_qubits, uint32_t init_index, const double* g_ao,
                                   const double* h_ao, const double* C, const double* nuc, int N,
                                   int n_occ, int ncas, const int32_t* kap_row, const int32_t* kap_col,
                                   int n_kappa, int derivatives, int batch, double* work, double* out,
                                   unsigned eri_flags, const double* g_packed, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(nuc, "oo_eval_batch: null nuc");
    return oo_eval_batched(theta, n_theta, gates, n_gates, n_qubits, init_index, g_ao, h_ao, C, 0.0,
                           nuc, N, n_occ, ncas, kap_row, kap_col, n_kappa, derivatives, batch, work,
                           out, eri_flags, stream, g_packed);
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
