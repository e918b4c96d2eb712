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
// The factorisation is right-looking over 16-column panels.  The panel being factored lives in LDS (two
// buffers: the trailing update writes the next panel's column straight into the other buffer, so the
// per-panel chain never goes through memory); the rest of the trailing matrix is a working copy in global
// memory that every wave only re-reads where it wrote itself (tile (i, j) belongs to wave (i mod 4, j mod 2)
// for the whole factorisation).  Per panel:
//   D  (wave 0)  Cholesky of the 16 x 16 diagonal tile in registers, lane c <-> column c, the whole tile kept
//                symmetric so that every operand of a step is either the lane's own register or a
//                `v_readlane` of the pivot lane: no LDS round trip inside the 16 steps;
//   S  (a thread per row below the tile)  x = a L_kk^-T by forward substitution against broadcast reads of L_kk;
//   U  (all waves)  C_ij -= L_i L_j^T on the fp64 matrix cores (K = 16: four v_mfma_f64_16x16x4 per tile, the
//                operand fragments two ds_read_b128 each from the panel buffer, pitch 18 doubles:
//                conflict-free).
// The factor overwrites the working copy's lower triangle; the back substitution reads its block rows from
// there (column-oriented: thread c owns y[c], no reductions), the 16 x 16 transposed solves again by readlane.
#include "common.h"
#include <math.h>

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

constexpr int CT = 512;              // threads per workgroup: 8 waves, two per SIMD (256 registers each)
constexpr int CWV = CT / 64;
constexpr int CP = 18;               // pitch of a panel row in LDS (doubles): 16-byte aligned rows, conflict-free fragments
constexpr int NCHOL_MAX = 495;       // n + 1 <= 496 rows = 31 tiles: two panel buffers of 496 x 18 doubles = 143 KB
constexpr int C_SC1 = 16;            // buffer aux bit sc1: loads bypass the L1

typedef unsigned c_v2u __attribute__((ext_vector_type(2)));

struct CholLds { int P0, P1, yv, xb, flag, total; };

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
    const int T = (n + 1 + 15) >> 4, npv = 16 * T, lda = npv;
    const double sigma = role == 0 ? 0.0 : lambda_min;
    const double* Hb = H + (size_t)prob * n * n;
    const double* gb = g + (size_t)prob * n;
    double* Aw = work + (size_t)blockIdx.x * npv * npv;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Aw, 0, npv * npv * 8, 0x00020000);
    double* P0 = sm + L.P0;
    double* P1 = sm + L.P1;
    double* yv = sm + L.yv;
    double* xb = sm + L.xb;
    int* flag = reinterpret_cast<int*>(sm + L.flag);

    // element (r, c) of the matrix being factored: [[H - sigma I, -g], [-g^T, 1]] padded with an identity
    // (the lower triangle of H is what eigh reads in the reference: torch.linalg.eigh, UPLO = 'L').
    // Branch-free: both loads are issued whatever (r, c) is (clamped addresses) and combined with 0 / 1
    // weights -- a select on a loaded value is compiled to a branch around the load, and a staging pass whose
    // loads wait for each other one by one costs more than the whole factorisation.  (A NaN or an Inf in H
    // or g may therefore leak into the padding: the factorisation then fails, as it must.)
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
    const int wi = wave >> 1, wj = wave & 1;        // the wave owns the tiles (i, j) with i % 4 == wi, j % 2 == wj
    const int prow = c_pi(lr);
    if (tid == 0) flag[0] = 0;
    for (int idx = tid; idx < npv; idx += CT) yv[idx] = 0.0;
    // ---- staging: every wave brings its own tiles of the lower block triangle in, lane (lr, lq) <-> row
    // 16 i + lr, columns 16 j + 4 lq .. + 3 (the layout of phase U): column 0 into the first panel buffer, the
    // others into the working copy.  Two passes, each straight-line per batch (so that all loads of a batch
    // are in flight together): interior tiles (0 < j < i < T - 1) are plain H, two 16-byte loads, eight tiles
    // per batch; the others (column 0, the diagonal, the last tile row) go element by element.
    const unsigned lofs_g = (unsigned)(lr * lda + 4 * lq);          // lane part of a tile's offset in the working copy
    {
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Hb), 0, n * n * 8, 0x00020000);
        const unsigned lofs_h = (unsigned)(lr * n + 4 * lq);
        const int js = wj == 0 ? 2 : 1;             // first column >= 1 of this wave
        int ti = wi, tj = js;
        while (ti < T - 1 && tj >= ti) { ti += 4; tj = js; }
        while (ti < T - 1) {
            unsigned ho[8], go[8];
            d2 lo[8], hi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = ti < T - 1;
                ho[u] = live ? (unsigned)(16 * ti * n + 16 * tj) + lofs_h : 0x10000000u;
                go[u] = live ? (unsigned)(16 * ti * lda + 16 * tj) + lofs_g : 0x10000000u;
                if (live) {
                    tj += 2;
                    while (ti < T - 1 && tj >= ti) { ti += 4; tj = js; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                lo[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rh, ho[u] * 8u, 0, 0));
                hi[u] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rh, ho[u] * 8u + 16u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c_st2(ra, go[u], lo[u]);
                c_st2(ra, go[u] + 2, hi[u]);
            }
        }
        // the other tiles of this wave: (i, 0), (i, i), (T - 1, j)
        for (int i = wi; i < T; i += 4)
            for (int j = wj; j <= i; j += 2) {
                if (!(j == 0 || j == i || i == T - 1)) continue;
                d4 c4;
#pragma unroll
                for (int e = 0; e < 4; ++e) c4[e] = elem(16 * i + lr, 16 * j + 4 * lq + e);
                d2 l2, h2; l2.x = c4[0]; l2.y = c4[1]; h2.x = c4[2]; h2.y = c4[3];
                if (j == 0) {
                    double* np = P0 + (16 * i + lr) * CP + 4 * lq;
                    *reinterpret_cast<d2*>(np) = l2;
                    *reinterpret_cast<d2*>(np + 2) = h2;
                } else {
                    const unsigned ge = (unsigned)(16 * i * lda + 16 * j) + lofs_g;
                    c_st2(ra, ge, l2);
                    c_st2(ra, ge + 2, h2);
                }
            }
    }
    __syncthreads();
    CH_MARK(0);

    for (int k = 0; k < T; ++k) {
        double* cur = (k & 1) ? P1 : P0;            // rows 16 k .. npv - 1 of the panel, buffer row 0 = matrix row 16 k
        double* nxt = (k & 1) ? P0 : P1;
        // ---- D: the diagonal tile -> X = L_kk^-1 in the head of cur
        if (wave == 0) {
#ifdef OOVQE_CHOL_PROBE
            // (ablation builds compute wrong numbers: no pivot is tested, so that the run keeps its full length)
            const bool ok = chol_diag16(cur, yv, 16 * k, 0, lane, Aw + (size_t)16 * k * lda + 16 * k, lda);
#else
            const bool ok = chol_diag16(cur, yv, 16 * k, n, lane, Aw + (size_t)16 * k * lda + 16 * k, lda);
#endif
            if (!ok && lane == 0) flag[0] = 1;
        }
        CH_MARK(1);
        __syncthreads();
        CH_MARK(2);
        if (flag[0]) break;
        // ---- S: the row tiles below, L_i = A_i X^T on the matrix cores (PI-permuted: the lane ends up with
        // four adjacent columns of its row), written back in place and into the working copy
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
                const unsigned ge = (unsigned)((16 * i + lr) * lda + 16 * k + 4 * lq);
                c_st2(ra, ge, lo);
                c_st2(ra, ge + 2, hi);
            }
        }
        CH_MARK(3);
        __syncthreads();
        CH_MARK(4);
        if (tid < 16 && n >= 16 * (k + 1) && 16 * k + tid < n)     // y rides along as row n of the factor
            yv[16 * k + tid] = cur[(n - 16 * k) * CP + tid];
        // ---- U: trailing tiles of this wave.  C_ij -= L_i L_j^T with the PI-permuted A operand (c_pi above):
        // the lane holds C[16 i + lr][16 j + 4 lq .. + 3].  Batches of four tiles, straight-line (the tail of a
        // pass is filled with tiles outside the matrix: their loads return zeros and their stores are dropped
        // by the descriptor's range check), so that the compiler's vmcnt bookkeeping stays exact and the loads
        // of the next batch are in flight under the products of the current one; inside a batch the fragments
        // of all four tiles are read first and the four accumulator chains are issued interleaved.
        struct Tile { unsigned g; int la, lb, ln; };     // offsets: working copy; A / B fragment rows; next panel
        auto mk = [&](int i, int j) -> Tile {
            Tile t;
            t.g = i < T ? (unsigned)(16 * i * lda + 16 * j) + lofs_g : 0x10000000u;
            t.la = 16 * (j - k) * CP;
            t.lb = 16 * (i - k) * CP;
            t.ln = 16 * (i - k - 1) * CP;
            return t;
        };
        const double* fa_base = cur + prow * CP + 4 * lq;
        const double* fb_base = cur + lr * CP + 4 * lq;
        double* nx_base = nxt + lr * CP + 4 * lq;
        auto gload4 = [&](const Tile (&t)[4], d4 (&cc)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned ge = t[u].g;
#if defined(OOVQE_CHOL_PROBE) && (OOVQE_CHOL_PROBE & 2)
                ge = 0x10000000u;
#endif
                const d2 lo = c_ld2(ra, ge), hi = c_ld2(ra, ge + 2);
                cc[u][0] = lo.x; cc[u][1] = lo.y; cc[u][2] = hi.x; cc[u][3] = hi.y;
            }
        };
        auto product4 = [&](const Tile (&t)[4], d4 (&cc)[4]) {
            d2 a01[4], a23[4], b01[4], b23[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a01[u] = *reinterpret_cast<const d2*>(fa_base + t[u].la);
                a23[u] = *reinterpret_cast<const d2*>(fa_base + t[u].la + 2);
                b01[u] = *reinterpret_cast<const d2*>(fb_base + t[u].lb);
                b23[u] = *reinterpret_cast<const d2*>(fb_base + t[u].lb + 2);
            }
#if defined(OOVQE_CHOL_PROBE) && (OOVQE_CHOL_PROBE & 4)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cc[u][0] += a01[u].x * b01[u].x; cc[u][1] += a01[u].y * b01[u].y;
                cc[u][2] += a23[u].x * b23[u].x; cc[u][3] += a23[u].y * b23[u].y;
            }
#else
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = mfma_f64(-a01[u].x, b01[u].x, cc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = mfma_f64(-a01[u].y, b01[u].y, cc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = mfma_f64(-a23[u].x, b23[u].x, cc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = mfma_f64(-a23[u].y, b23[u].y, cc[u]);
#endif
        };
        auto gstore4 = [&](const Tile (&t)[4], const d4 (&cc)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned ge = t[u].g;
#if defined(OOVQE_CHOL_PROBE) && (OOVQE_CHOL_PROBE & 1)
                ge = 0x10000000u;
#endif
                d2 lo, hi; lo.x = cc[u][0]; lo.y = cc[u][1]; hi.x = cc[u][2]; hi.y = cc[u][3];
                c_st2(ra, ge, lo);
                c_st2(ra, ge + 2, hi);
            }
        };
        int i_first = k + 1;
        while ((i_first & 3) != wi) ++i_first;
        {
            // pass A: the next panel's column (j = k + 1) first -> LDS
            if (wj == ((k + 1) & 1)) {
                for (int i0 = i_first; i0 < T; i0 += 16) {
                    Tile ta[4];
                    d4 ca[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ta[u] = mk(i0 + 4 * u < T ? i0 + 4 * u : T, k + 1);
                    gload4(ta, ca);
                    product4(ta, ca);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        d2 lo, hi; lo.x = ca[u][0]; lo.y = ca[u][1]; hi.x = ca[u][2]; hi.y = ca[u][3];
                        *reinterpret_cast<d2*>(nx_base + ta[u].ln) = lo;
                        *reinterpret_cast<d2*>(nx_base + ta[u].ln + 2) = hi;
                    }
                }
            }
            // pass B: columns j >= k + 2 -> working copy, two batches of four in flight
            const int j0 = ((k + 2) & 1) == wj ? k + 2 : k + 3;
            int ti = i_first, tj = j0;
            while (ti < T && tj > ti) { ti += 4; tj = j0; }
            auto gather = [&](Tile (&t)[4]) -> bool {
                const bool any = ti < T;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    t[u] = mk(ti < T ? ti : T, ti < T ? tj : T);
                    if (ti < T) {
                        tj += 2;
                        while (ti < T && tj > ti) { ti += 4; tj = j0; }
                    }
                }
                return any;
            };
            Tile ta[4], tb[4];
            d4 ca[4], cb[4];
            bool more = gather(ta);
            gload4(ta, ca);
            while (more) {
                const bool more_b = gather(tb);
                gload4(tb, cb);
                product4(ta, ca);
                gstore4(ta, ca);
                if (!more_b) break;
                more = gather(ta);
                gload4(ta, ca);
                product4(tb, cb);
                gstore4(tb, cb);
            }
        }
        CH_MARK(5);
        __syncthreads();
        CH_MARK(6);
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
            lrow[r] = (tid < 16 * kb) ? c_ld1(ra, (unsigned)((16 * kb + r) * lda + tid)) : 0.0;
    };
    auto fetch_diag = [&](int kb) {                 // lane c of wave 0: column c of X = L_kk^-1
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ldg[r] = (wave == 0) ? c_ld1(ra, (unsigned)((16 * kb + r) * lda + 16 * kb + lr)) : 0.0;
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

// info[b] = 1 when both roles of problem b succeeded (dp[b] final, shift[b] = 0), else 0
__global__ void newton_chol_info_kernel(const int* __restrict__ status, double* __restrict__ info, int batch, int roles)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    bool ok = true;
    for (int r = 0; r < roles; ++r) ok = ok && status[b * roles + r] == 1;
    info[b] = ok ? 1.0 : 0.0;
}

}  // namespace

int oovqe_newton_chol_max_n(void) { return NCHOL_MAX; }

// doubles of workspace: the working copies of both roles of every problem + their status words
size_t oovqe_newton_chol_work(int n, int batch)
{
    const size_t npv = 16 * (size_t)((n + 1 + 15) / 16);
    return (size_t)batch * 2 * npv * npv + (size_t)batch + 2;
}

int oovqe_newton_chol_launch(const double* hessian, const double* gradient, int n, int batch, double lambda_min,
                             double* work, double* dp, double* shift, double* info, hipStream_t st)
{
    OOVQE_REQUIRE(n >= 1 && n <= NCHOL_MAX, "newton_direction_pd: n = %d outside 1..%d", n, NCHOL_MAX);
    const CholLds L = chol_lds(n);
    const size_t lds = (size_t)L.total * sizeof(double);
    OOVQE_REQUIRE(lds <= 160 * 1024, "newton_direction_pd: %zu bytes of LDS needed", lds);
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)newton_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds),
                    "newton_direction_pd: hipFuncSetAttribute");
    const int roles = lambda_min != 0.0 ? 2 : 1;
    const size_t npv = 16 * (size_t)((n + 1 + 15) / 16);
    int* status = reinterpret_cast<int*>(work + (size_t)batch * 2 * npv * npv);
    hipLaunchKernelGGL(newton_chol_kernel, dim3(batch * roles), dim3(CT), lds, st, hessian, gradient, n, lambda_min,
                       work, dp, shift, status, roles);
    OOVQE_CHECK_LAUNCH("newton_direction_pd");
    hipLaunchKernelGGL(newton_chol_info_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, status, info, batch, roles);
    OOVQE_CHECK_LAUNCH("newton_direction_pd/info");
    return 0;
}
