// Newton direction when the Hessian is positive definite: blocked Cholesky, one workgroup per problem.
//
// What the reference computes (src/auto_oo/utils/newton_raphson.py:105-128): `eigh` of the Hessian; only when
// the lowest eigenvalue is below lambda_min is the Hessian shifted; then dp = -H^-1 g through the
// eigendecomposition.  So when lambda_low >= lambda_min the direction does not depend on the spectrum at all:
// it is -H^-1 g, and lambda_low is just a reported number.  The band-reduction route of newton.hip pays
// ~1.1 ms of latency chain (42 panels x two inter-workgroup hand-offs, then 4-5 multisection rounds) before
// it can solve; here the same direction comes from a Cholesky factorisation whose critical path has no
// hand-off between workgroups at all:
//
//   role 0 (workgroup 2 b):      H = L L^T, forward substitution fused into the factorisation (-g rides along
//                                as row n of the matrix: L[n][0..n-1] = y with L y = -g), back substitution,
//                                dp = x.  Succeeds  <=>  H is (numerically) positive definite.
//   role 1 (workgroup 2 b + 1):  the same factorisation of H - lambda_min I, WITHOUT the solve: succeeds <=>
//                                lambda_low > lambda_min, i.e. the reference would not shift.
//   both succeed -> info[b] = 1: dp is final, shift = 0.  Otherwise info[b] = 0 and the band route of
//   newton.hip (lowest eigenvalue, level shift, band solve) computes the direction as before.
//   The two workgroups never wait for each other: nothing here needs co-residency.
//
// The factorisation is LEFT-looking over 16-column panels with one panel of look-ahead (the first version
// was right-looking: its trailing update re-read and re-wrote the whole working copy once per panel and was
// bound by that traffic, 320 us; this one 187 us at n = 331).  Six waves own the row tiles of a panel and
// keep them in registers from the first contribution to the last; the factor is only ever appended to.
// Per panel k:
//   1  (the six accumulating waves)  the last contribution, p = k - 1, from the previous panel's buffer in LDS,
//      plus the matrix' own tile; panel k goes into its LDS buffer (two buffers, alternating);
//   2  (wave 0)  D: Cholesky of the 16 x 16 diagonal tile AND the inverse X of its factor, in registers, lane c
//      <-> column c, the tile kept symmetric so that every operand of a step is either the lane's own register
//      or a `v_readlane` of the pivot lane -- no LDS round trip inside the 16 steps; square-root-free
//      recurrences, so that the chain from pivot to pivot is readlane -> reciprocal -> multiply -> fma;
//      (the others, meanwhile)  panel k + 1 already: acc = - sum_{p < k} L_ip L_k+1,p^T on the fp64 matrix cores,
//      both operands straight from the factor in memory, which is stored tile by tile in FRAGMENT order (a
//      wave's 16-byte load = one contiguous KB; row-major tiles streamed at a sixth of the rate), the next
//      panel's operands in flight under the products of the current one;
//   3  (all)  S: the row tiles below the diagonal, L_i = A_i X^T as one matrix-core product each (the inverse
//      instead of a substitution: a thread per row spent its time on broadcast reads of L_kk), written back
//      into the panel buffer and appended to the factor (both operand orders).
// The matrix cores deliver D[m][n] to lane (n, lq) as rows m = lq + 4 e; with the A operand's rows permuted
// (c_pi) the lane holds four ADJACENT columns of one row: 16-byte accesses everywhere.
// Back substitution: column-oriented (thread c owns y[c], no reductions), the 16 x 16 transposed solves are
// mat-vecs with the stored inverses, y of a block by readlane.
#include "common.h"
#include <math.h>
#include <type_traits>

int oovqe_newton_chol_launch(const double* hessian, const double* gradient, int n, int batch, double lambda_min,
                             double* work, double* dp, double* shift, double* info, hipStream_t st);
size_t oovqe_newton_chol_work(int n, int batch);
int oovqe_newton_chol_max_n(void);

#ifdef OOVQE_CHOL_TIMING
// tools/newton_chol_probe.hip: cycles per phase, thread 0 of workgroup 0
__device__ long long g_chol_cycles[16];
#define CH_MARK(k)                                                                     \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                     \
            const long long now__ = clock64();                                         \
            g_chol_cycles[k] += now__ - t_mark;                                        \
            t_mark = now__;                                                            \
        }                                                                              \
    } while (0)
#else
#define CH_MARK(k) do {} while (0)
#endif

namespace {

#ifndef OOVQE_CHOL_CT
#define OOVQE_CHOL_CT 512
#endif
constexpr int CT = OOVQE_CHOL_CT;    // threads per workgroup: 8 waves, two per SIMD (256 registers each) -- or 16, four per SIMD
constexpr int CWV = CT / 64;
constexpr int WJ = CWV / 4;          // tile (i, j) belongs to wave (i % 4) * WJ + j % WJ
constexpr int NBT = CT == 512 ? 4 : 2;   // trailing tiles per batch of a wave (two batches in flight)
constexpr int NST = CT == 512 ? 8 : 4;   // interior tiles per staging batch
constexpr int CP = 18;               // pitch of a panel row in LDS (doubles): 16-byte aligned rows, conflict-free fragments
constexpr int NCHOL_MAX = 495;       // n + 1 <= 496 rows = 31 tiles: two panel buffers of 496 x 18 doubles = 143 KB
constexpr int C_SC1 = 16;            // buffer aux bit sc1: loads bypass the L1

typedef unsigned c_v2u __attribute__((ext_vector_type(2)));

struct CholLds { int P0, P1, yv, xb, flag, total; };

// doubles of one factorisation's workspace: La, Lb (T x T tiles of 256) and Xw (T tiles)
__host__ __device__ inline size_t chol_factor_doubles(int T) { return (size_t)(2 * T * T + T) * 256; }

__host__ __device__ inline CholLds chol_lds(int n)
{
    const int npv = 16 * ((n + 1 + 15) / 16);
    CholLds L;
    int o = 0;
    L.P0 = o; o += npv * CP;
    L.P1 = o; o += npv * CP;
    L.yv = o; o += npv;
    L.xb = o; o += 16;
    L.flag = o; o += 2;
    L.total = o;
    return L;
}

__device__ __forceinline__ double c_lane(double x, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src),
                            __builtin_amdgcn_readlane(__double2loint(x), src));
}

// 1 / sqrt(p) to rounding accuracy: hardware estimate + two coupled Newton (Goldschmidt) steps
__device__ __forceinline__ double c_rsqrt(double p)
{
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return h + h;
}

__device__ __forceinline__ double c_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double c_ld1(__amdgpu_buffer_rsrc_t r, unsigned elem)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, elem * 8u, 0, C_SC1));
}
__device__ __forceinline__ void c_st1(__amdgpu_buffer_rsrc_t r, unsigned elem, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(c_v2u, v), r, elem * 8u, 0, 0);
}

// The matrix cores deliver D[m][n] to lane (lr = n, lq) as the four rows m = lq + 4 e.  Feeding the A operand
// with its rows permuted by PI turns that into four ADJACENT columns of one matrix row: with
// A[m][k] = L_j[PI(m)][k] and B[k][n] = L_i[n][k] the lane holds C[16 i + lr][16 j + 4 lq + e], e = 0..3 --
// 32 contiguous bytes, two 16-byte accesses per tile instead of four 8-byte ones.
__device__ __forceinline__ int c_pi(int m) { return 4 * (m & 3) + (m >> 2); }

__device__ __forceinline__ d2 c_ld2(__amdgpu_buffer_rsrc_t r, unsigned elem)
{
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, elem * 8u, 0, C_SC1));
}
typedef unsigned c_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void c_st2(__amdgpu_buffer_rsrc_t r, unsigned elem, d2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c_v4u, v), r, elem * 8u, 0, 0);
}

// Cholesky of the 16 x 16 diagonal tile at the head of `cur` (rows 0..15, pitch CP) AND the inverse of its
// factor, by one wave.  Lane c (c = lane & 15; the four 16-lane rows of the wave carry copies) holds column c
// of the tile, which is kept symmetric: step j needs the pivot column's entries (readlane of lane j: the
// same number in every lane) and A[j][c] = A[c][j] (the lane's own register j) -- no LDS round trip inside
// the 16 steps.  The inverse X = L_kk^-1 rides on the same uniform numbers: X[j][c] = -s_j sum_{k<j} L[j][k]
// X[k][c], accumulated right-looking (acc[r] += L[r][j] X[j][c] at step j), so that the rows below the tile
// become one matrix-core product with X (phase S) and the back substitution's 16 x 16 solves mat-vecs.  The
// pivot of step j + 1 and its reciprocal square root are formed first, the other updates of step j run in
// the shadow of that chain.  On return cur holds X (row j, column c; zero above the diagonal), and so does
// the tile's place in the working copy.  Pivots of rows >= n (the right-hand-side row and the padding) are
// not tested and count as 1.  Returns false on a pivot <= 0 (or NaN).
__device__ __forceinline__ bool chol_diag16(double* __restrict__ cur, double* __restrict__ yv, int row0, int n,
                                            int lane, double* __restrict__ awd, int lda)
{
    int c = lane & 15;
    // (opaque to the optimiser: otherwise the sixteen (c == j) selects below are hoisted out of the panel
    // loop as loop invariants and live -- spilled -- across the whole kernel)
    asm volatile("" : "+v"(c));
    double a[16], acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = cur[r * CP + c]; acc[r] = 0.0; }
    const bool rhs_row = row0 + c == n;              // the right-hand-side row ends inside this tile
    // The recurrences run in the square-root-free form (A = Lu D Lu^T, Lu unit lower): the chain from one
    // pivot to the next is readlane -> reciprocal -> one multiply -> one fma -> readlane; the reciprocal square
    // root of a pivot only scales what leaves the tile (row j of X = rsqrt(d_j) x row j of Lu^-1, L[c][j] =
    // A_j[j][c] rsqrt(d_j)) and is off that chain.
    bool ok = true;
    double p = c_lane(a[0], 0);
    double rd = c_rcp(row0 < n ? p : 1.0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // (the steps are kept apart for the instruction scheduler: let loose on all 16 of them it hoists the
        // readlanes of later steps and spills both register files)
        __builtin_amdgcn_sched_barrier(0);
        const bool live = row0 + j < n;
        ok = ok && (!live || p > 0.0);
        const double xu = (c == j ? 1.0 : 0.0) - acc[j];     // Lu^-1[j][c]; the sum is 0 by itself for c >= j
        const double aj = a[j];                               // A_j[j][c] = d_j Lu[c][j]
        double pn = 1.0, rdn = 1.0;
        if (j + 1 < 16) {
            const double l1 = c_lane(a[j + 1], j) * rd;
            a[j + 1] = fma(-l1, aj, a[j + 1]);
            acc[j + 1] = fma(l1, xu, acc[j + 1]);
            pn = c_lane(a[j + 1], j + 1);
            rdn = c_rcp(row0 + j + 1 < n ? pn : 1.0);
        }
        const double sj = c_rsqrt(live ? p : 1.0);
        const double xj = sj * xu;
        // row j of X is final: out it goes (the tile's own entries were all read before the first step; the
        // four 16-lane copies write the same numbers)
        cur[j * CP + c] = xj;
        awd[(size_t)j * lda + c] = xj;
        if (rhs_row && j < c) yv[row0 + j] = aj * sj;
#pragma unroll
        for (int r = j + 2; r < 16; ++r) {
            const double lr = c_lane(a[r], j) * rd;  // Lu[r][j], the same number in every lane
            a[r] = fma(-lr, aj, a[r]);
            acc[r] = fma(lr, xu, acc[r]);
        }
        p = pn; rd = rdn;
    }
    __builtin_amdgcn_sched_barrier(0);
    return ok;
}

// NTW: row tiles of a panel per accumulating wave: ceil(T / 6) <= NTW; DEPTH: operand panels in flight
template <int NTW, int DEPTH>
__global__ __launch_bounds__(CT)
void newton_chol_kernel(const double* __restrict__ H, const double* __restrict__ g, int n, double lambda_min,
                        double* __restrict__ work, double* __restrict__ dp, double* __restrict__ shift,
                        int* __restrict__ status, int roles)
{
    extern __shared__ double sm[];
    const CholLds L = chol_lds(n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int prob = blockIdx.x / roles, role = blockIdx.x - prob * roles;
    const int T = (n + 1 + 15) >> 4, npv = 16 * T;
    const double sigma = role == 0 ? 0.0 : lambda_min;
    const double* Hb = H + (size_t)prob * n * n;
    const double* gb = g + (size_t)prob * n;
    // The factor lives in memory tile by tile in FRAGMENT order, twice: in tile (i, p) of Lb the 32 bytes of
    // lane l = lr + 16 lq are row lr, columns 4 lq .. + 3 of L_ip (the B operand and the accumulator layout), in
    // La they are row PI(lr) (the A operand).  Either way a wave's 16-byte load is one contiguous KB -- with
    // row-major tiles the 64 lanes of a load touch 16 cache lines, four lanes each, and the factor streamed at
    // a sixth of the rate.  Xw: the inverses of the diagonal tiles, row-major (the back substitution).
    const size_t fsz = chol_factor_doubles(T);
    double* Aw = work + (size_t)blockIdx.x * fsz;
    const unsigned la_off = 0u, lb_off = (unsigned)(T * T * 256), xw_off = 2u * (unsigned)(T * T * 256);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Aw, 0, (int)(fsz * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Hb), 0, n * n * 8, 0x00020000);
    double* P0 = sm + L.P0;
    double* P1 = sm + L.P1;
    double* yv = sm + L.yv;
    double* xb = sm + L.xb;
    int* flag = reinterpret_cast<int*>(sm + L.flag);

    // element (r, c) of the matrix being factored: [[H - sigma I, -g], [-g^T, 1]] padded with an identity
    // (the lower triangle of H is what eigh reads in the reference: torch.linalg.eigh, UPLO = 'L').
    // Branch-free: both loads are issued whatever (r, c) is (clamped addresses) and combined with 0 / 1
    // weights -- a select on a loaded value is compiled to a branch around the load, and loads that wait for
    // each other one by one cost more than the whole factorisation.  (A NaN or an Inf in H or g may
    // therefore leak into the padding: the factorisation then fails, as it must.)
    const int nm1 = n - 1;
    auto elem = [&](int r, int c) -> double {
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        const int hc = hi < nm1 ? hi : nm1, lc = lo < nm1 ? lo : nm1;
        const double vh = Hb[(size_t)hc * n + lc];
        const double vg = gb[lc];
        const double mh = hi < n ? 1.0 : 0.0;
        const double mg = (hi == n && lo < n) ? 1.0 : 0.0;
        const double md = r == c ? (hi < n ? -sigma : 1.0) : 0.0;
        return fma(mh, vh, fma(-mg, vg, md));
    };

#ifdef OOVQE_CHOL_TIMING
    long long t_mark = clock64();
#endif
    const int prow = c_pi(lr);
    const unsigned lofs = (unsigned)(2 * lane);                 // the lane's 16 bytes in a half tile of the factor
    const unsigned lofs_p = (unsigned)(2 * (prow + 16 * lq));   // ... of the lane that holds this lane's row in La
    const unsigned lofs_h = (unsigned)(lr * n + 4 * lq);
    constexpr unsigned NOWHERE = 0x10000000u;                   // an element offset beyond every buffer: loads give 0, stores are dropped
    if (tid == 0) flag[0] = 0;
    for (int idx = tid; idx < npv; idx += CT) yv[idx] = 0.0;

    // Left-looking over 16-column panels, one panel of look-ahead.  Six waves own the row tiles of a panel
    // (tile i of panel kp belongs to accumulating wave (i - kp) % 6) and keep them in registers from the first
    // contribution to the last: acc = - sum_{p < kp} L_ip L_kp,p^T, with both operands straight from the factor
    // in memory in fragment layout (two 16-byte loads per operand and panel p, the next p in flight under
    // the products of the current one; the A operand is the same for all waves: L1 hits).  While wave 0
    // factors the diagonal tile of panel k (phase 2), the others already accumulate panel k + 1 over p < k;
    // the last contribution (p = k) comes from the panel buffer in LDS once phase 3 has produced it.
    // (wave 4 shares its SIMD with wave 0 -- a workgroup's waves go round the four SIMDs -- and every matrix-core
    // instruction it issued would hold up the vector instructions of the diagonal-tile factorisation: it sits
    // the accumulation out, like wave 0)
    constexpr int NBW = CWV - 2;                    // accumulating waves
    const int bw = wave == 0 || wave == 4 ? -1 : (wave < 4 ? wave - 1 : wave - 2);
    auto tile_of = [&](int kp, int u) -> int { return bw < 0 ? T : kp + bw + NBW * u; };
    d4 acc[NTW], atile[NTW];
    auto load_a = [&](int kp) {                     // the tiles (i, kp) of the matrix itself
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const int i = tile_of(kp, u);
            if (i < T - 1 && i != kp) {
                const unsigned he = ((unsigned)(16 * i * n + 16 * kp) + lofs_h) * 8u;
                const d2 lo = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rh, he, 0, 0));
                const d2 hi = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rh, he + 16u, 0, 0));
                atile[u][0] = lo.x; atile[u][1] = lo.y; atile[u][2] = hi.x; atile[u][3] = hi.y;
            } else if (i < T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atile[u][e] = elem(16 * i + lr, 16 * kp + 4 * lq + e);
            } else {
                atile[u] = d4{0.0, 0.0, 0.0, 0.0};
            }
        }
    };
    // acc -= sum_{p < pend} L_ip L_kp,p^T for the NT live tiles of this wave in panel kp (NT a compile-time
    // constant: a tile slot beyond the matrix would cost its four products per panel all the same)
    auto bulk = [&](auto nt_c, int kp, int pend) {
        constexpr int NT = decltype(nt_c)::value;
        struct Frag { d2 a01, a23, b01[NT], b23[NT]; };
        const unsigned ao = la_off + (unsigned)(kp * T * 256) + lofs;
        unsigned bo[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) bo[u] = lb_off + (unsigned)(tile_of(kp, u) * T * 256) + lofs;
        auto fetch = [&](Frag& f, int p) {
#if defined(OOVQE_CHOL_PROBE) && (OOVQE_CHOL_PROBE & 2)
            const unsigned po = NOWHERE + (unsigned)(p - p);
#else
            const unsigned po = p < pend ? (unsigned)(256 * p) : NOWHERE;
#endif
            f.a01 = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(ra, (ao + po) * 8u, 0, 0));
            f.a23 = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(ra, (ao + po) * 8u + 1024u, 0, 0));
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f.b01[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(ra, (bo[u] + po) * 8u, 0, 0));
                f.b23[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(ra, (bo[u] + po) * 8u + 1024u, 0, 0));
            }
        };
        auto mma = [&](const Frag& f) {
#if defined(OOVQE_CHOL_PROBE) && (OOVQE_CHOL_PROBE & 4)
#pragma unroll
            for (int u = 0; u < NT; ++u) { acc[u][0] += f.a01.x * f.b01[u].x; acc[u][1] += f.a01.y * f.b01[u].y; acc[u][2] += f.a23.x * f.b23[u].x; acc[u][3] += f.a23.y * f.b23[u].y; }
            return;
#endif
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = mfma_f64(-f.a01.x, f.b01[u].x, acc[u]);
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = mfma_f64(-f.a01.y, f.b01[u].y, acc[u]);
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = mfma_f64(-f.a23.x, f.b23[u].x, acc[u]);
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = mfma_f64(-f.a23.y, f.b23[u].y, acc[u]);
        };
        // DEPTH - 1 panels of operands in flight per wave; panels beyond pend are fetched from nowhere (zeros) and
        // never multiplied
        Frag f[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j) fetch(f[j], j);
        for (int p = 0; p < pend; p += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                fetch(f[(j + DEPTH - 1) % DEPTH], p + j + DEPTH - 1);
                if (p + j < pend) mma(f[j]);
            }
        }
    };
    auto bulk_live = [&](int kp, int pend) {        // dispatch on the number of live tiles of this wave in panel kp
        const int live = bw < 0 || kp + bw >= T ? 0 : (T - 1 - kp - bw) / NBW + 1;
        if (live == 1) bulk(std::integral_constant<int, 1>{}, kp, pend);
        else if (live == 2) bulk(std::integral_constant<int, 2>{}, kp, pend);
        else if (live == 3) bulk(std::integral_constant<int, (NTW >= 3 ? 3 : 1)>{}, kp, pend);
        else if (live == 4) bulk(std::integral_constant<int, (NTW >= 4 ? 4 : 1)>{}, kp, pend);
        else if (live >= 5) bulk(std::integral_constant<int, NTW>{}, kp, pend);
    };

#pragma unroll
    for (int u = 0; u < NTW; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
    load_a(0);
    CH_MARK(0);

    for (int k = 0; k < T; ++k) {
        double* cur = (k & 1) ? P1 : P0;            // rows 16 k .. npv - 1 of the panel, buffer row 0 = matrix row 16 k
        const double* prv = (k & 1) ? P0 : P1;      // panel k - 1: buffer row 0 = matrix row 16 (k - 1)
        // ---- phase 1: the last contribution (p = k - 1, from the panel buffer), the matrix' own tile, and
        // panel k goes into its buffer
        if (bw >= 0) {
            const double* ap = prv + (16 + prow) * CP + 4 * lq;            // L_k,k-1, rows permuted
            d2 a01 = {0.0, 0.0}, a23 = {0.0, 0.0};
            if (k > 0) { a01 = *reinterpret_cast<const d2*>(ap); a23 = *reinterpret_cast<const d2*>(ap + 2); }
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int i = tile_of(k, u);
                if (i < T) {
                    d4 c4 = acc[u];
                    if (k > 0) {
                        const double* bp = prv + (16 * (i - k + 1) + lr) * CP + 4 * lq;
                        const d2 b01 = *reinterpret_cast<const d2*>(bp), b23 = *reinterpret_cast<const d2*>(bp + 2);
                        c4 = mfma_f64(-a01.x, b01.x, c4);
                        c4 = mfma_f64(-a01.y, b01.y, c4);
                        c4 = mfma_f64(-a23.x, b23.x, c4);
                        c4 = mfma_f64(-a23.y, b23.y, c4);
                    }
                    c4 += atile[u];
                    d2 lo, hi; lo.x = c4[0]; lo.y = c4[1]; hi.x = c4[2]; hi.y = c4[3];
                    double* np = cur + (16 * (i - k) + lr) * CP + 4 * lq;
                    *reinterpret_cast<d2*>(np) = lo;
                    *reinterpret_cast<d2*>(np + 2) = hi;
                }
            }
        }
        CH_MARK(5);
        __syncthreads();
        CH_MARK(6);
        // ---- phase 2: wave 0 factors the diagonal tile (-> X = L_kk^-1 in the head of cur); the others start on
        // panel k + 1: its own tiles of the matrix and every contribution that is already final (p < k)
        if (wave == 0) {
#ifdef OOVQE_CHOL_PROBE
            // (ablation builds compute wrong numbers: no pivot is tested, so that the run keeps its full length)
            const bool ok = chol_diag16(cur, yv, 16 * k, 0, lane, Aw + xw_off + (size_t)k * 256, 16);
#else
            const bool ok = chol_diag16(cur, yv, 16 * k, n, lane, Aw + xw_off + (size_t)k * 256, 16);
#endif
            if (!ok && lane == 0) flag[0] = 1;
        } else if (k + 1 < T && bw >= 0) {
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
            load_a(k + 1);
            bulk_live(k + 1, k);
        }
        CH_MARK(1);
        __syncthreads();
        CH_MARK(2);
        if (flag[0]) break;
        // ---- phase 3: the row tiles below, L_i = A_i X^T on the matrix cores (PI-permuted: the lane ends up with
        // four adjacent columns of its row), written back in place and into the factor
        {
            const double* xp = cur + prow * CP + 4 * lq;
            const d2 x01 = *reinterpret_cast<const d2*>(xp), x23 = *reinterpret_cast<const d2*>(xp + 2);
            for (int i = k + 1 + wave; i < T; i += CWV) {
                double* bp = cur + (16 * (i - k) + lr) * CP + 4 * lq;
                const d2 b01 = *reinterpret_cast<const d2*>(bp), b23 = *reinterpret_cast<const d2*>(bp + 2);
                d4 c4 = {0.0, 0.0, 0.0, 0.0};
                c4 = mfma_f64(x01.x, b01.x, c4);
                c4 = mfma_f64(x01.y, b01.y, c4);
                c4 = mfma_f64(x23.x, b23.x, c4);
                c4 = mfma_f64(x23.y, b23.y, c4);
                d2 lo, hi;
                lo.x = c4[0]; lo.y = c4[1]; hi.x = c4[2]; hi.y = c4[3];
                *reinterpret_cast<d2*>(bp) = lo;
                *reinterpret_cast<d2*>(bp + 2) = hi;
                const unsigned tb = (unsigned)((i * T + k) * 256);
                c_st2(ra, lb_off + tb + lofs, lo);
                c_st2(ra, lb_off + tb + 128u + lofs, hi);
                c_st2(ra, la_off + tb + lofs_p, lo);
                c_st2(ra, la_off + tb + 128u + lofs_p, hi);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);              // (this panel of the factor is in memory before anybody reads it)
        CH_MARK(3);
        __syncthreads();
        CH_MARK(4);
        if (tid < 16 && n >= 16 * (k + 1) && 16 * k + tid < n)     // y rides along as row n of the factor
            yv[16 * k + tid] = cur[(n - 16 * k) * CP + tid];
    }

    const bool failed = flag[0] != 0;
    if (role != 0) {
        if (tid == 0) status[blockIdx.x] = failed ? 0 : 1;
        return;
    }
    if (failed) {
        if (tid == 0) status[blockIdx.x] = 0;
        return;
    }
    // ---- L^T x = y on the leading n x n block, block rows from the last one up.  Thread c owns y[c]; the
    // block row kb of L (16 rows x 16 kb columns, in the working copy) and the inverse of its diagonal tile
    // are fetched a block ahead; x_blk = X^T y_blk is a mat-vec (the y of a block by readlane).
    __builtin_amdgcn_s_waitcnt(0);                  // (the factor's stores of every wave are out)
    __syncthreads();
    const int nb = (n + 15) >> 4;
    double lrow[16], ldg[16];
    auto fetch_rows = [&](int kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            // element (row r, column c = tid) of block row kb: tile (kb, c / 16), slot r + 16 (c' / 4), half
            // (c' % 4) / 2, position c' % 2
            lrow[r] = (tid < 16 * kb) ? c_ld1(ra, lb_off + (unsigned)((kb * T + (tid >> 4)) * 256) +
                                                   (unsigned)((((tid & 15) & 3) >> 1) * 128 + 2 * (r + 16 * ((tid & 15) >> 2)) + (tid & 1)))
                                      : 0.0;
    };
    auto fetch_diag = [&](int kb) {                 // lane c of wave 0: column c of X = L_kk^-1
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ldg[r] = (wave == 0) ? c_ld1(ra, xw_off + (unsigned)(kb * 256 + r * 16 + lr)) : 0.0;
    };
    fetch_diag(nb - 1);
    fetch_rows(nb - 1);
    for (int kb = nb - 1; kb >= 0; --kb) {
        if (wave == 0) {
            const double yl = yv[16 * kb + lr];     // (rows >= n of the last block hold zeros)
            double x0 = 0.0, x1 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                x0 = fma(ldg[r], c_lane(yl, r), x0);
                x1 = fma(ldg[r + 1], c_lane(yl, r + 1), x1);
            }
            const double xfin = 16 * kb + lr < n ? x0 + x1 : 0.0;
            if (lane < 16) { xb[lane] = xfin; yv[16 * kb + lane] = xfin; }
        }
        CH_MARK(7);
        __syncthreads();
        CH_MARK(8);
        if (kb > 0) {
            if (tid < 16 * kb) {
                double acc = yv[tid];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc = fma(-lrow[r], xb[r], acc);
                yv[tid] = acc;
            }
            fetch_diag(kb - 1);
            fetch_rows(kb - 1);
        }
        CH_MARK(9);
        __syncthreads();
        CH_MARK(10);
    }
    for (int i2 = tid; i2 < n; i2 += CT) dp[(size_t)prob * n + i2] = yv[i2];
    if (tid == 0) {
        if (shift) shift[prob] = 0.0;
        status[blockIdx.x] = 1;
    }
}

// info[b] = 1 when both roles of problem b succeeded (dp[b] final, shift[b] = 0), else 0 -- and then dp[b] = 0: a
// caller that speculates (oovqe_oo_newton_step_batch: the band route of "the others" on a side stream) evaluates
// its first trial on whatever dp holds, and a zero direction is a trial at the current point, never a NaN orbital
// rotation.  One workgroup per problem.
__global__ void newton_chol_info_kernel(const int* __restrict__ status, double* __restrict__ info,
                                        double* __restrict__ dp, int n, int roles)
{
    const int b = blockIdx.x;
    bool ok = true;
    for (int r = 0; r < roles; ++r) ok = ok && status[b * roles + r] == 1;
    if (threadIdx.x == 0) info[b] = ok ? 1.0 : 0.0;
    if (!ok)
        for (int i = threadIdx.x; i < n; i += blockDim.x) dp[(size_t)b * n + i] = 0.0;
}

}  // namespace

int oovqe_newton_chol_max_n(void) { return NCHOL_MAX; }

// doubles of workspace: the working copies of both roles of every problem + their status words
size_t oovqe_newton_chol_work(int n, int batch)
{
    return (size_t)batch * 2 * chol_factor_doubles((n + 1 + 15) / 16) + (size_t)batch + 2;
}

int oovqe_newton_chol_launch(const double* hessian, const double* gradient, int n, int batch, double lambda_min,
                             double* work, double* dp, double* shift, double* info, hipStream_t st)
{
    OOVQE_REQUIRE(n >= 1 && n <= NCHOL_MAX, "newton_direction_pd: n = %d outside 1..%d", n, NCHOL_MAX);
    const CholLds L = chol_lds(n);
    const size_t lds = (size_t)L.total * sizeof(double);
    OOVQE_REQUIRE(lds <= 160 * 1024, "newton_direction_pd: %zu bytes of LDS needed", lds);
    const int T = (n + 1 + 15) / 16;
    const bool small = T <= 24;                     // four row tiles of a panel per accumulating wave (n <= 383), else six
    const void* kern = small ? (const void*)newton_chol_kernel<4, 2> : (const void*)newton_chol_kernel<6, 2>;
    {
        int rc_lds = oovqe_ensure_dynamic_lds(kern, lds);
        if (rc_lds) return rc_lds;
    }
    const int roles = lambda_min != 0.0 ? 2 : 1;
    int* status = reinterpret_cast<int*>(work + (size_t)batch * 2 * chol_factor_doubles(T));
    if (small)
        hipLaunchKernelGGL((newton_chol_kernel<4, 2>), dim3(batch * roles), dim3(CT), lds, st, hessian, gradient, n,
                           lambda_min, work, dp, shift, status, roles);
    else
        hipLaunchKernelGGL((newton_chol_kernel<6, 2>), dim3(batch * roles), dim3(CT), lds, st, hessian, gradient, n,
                           lambda_min, work, dp, shift, status, roles);
    OOVQE_CHECK_LAUNCH("newton_direction_pd");
    hipLaunchKernelGGL(newton_chol_info_kernel, dim3(batch), dim3(256), 0, st, status, info, dp, n, roles);
    OOVQE_CHECK_LAUNCH("newton_direction_pd/info");
    return 0;
}
