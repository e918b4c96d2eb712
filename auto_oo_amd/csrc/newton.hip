// Damped-Newton direction on the device: lowest Hessian eigenvalue, level shift and
// dp = -(H + nu I)^-1 g in ONE launch, one workgroup per problem of a batch.
//
// Replaces NewtonStep.newton_step of the reference (src/auto_oo/utils/newton_raphson.py:78-129:
// eigh of the (n_theta + n_kappa)^2 Hessian, a second eigh of the augmented Hessian when the
// lowest eigenvalue is below lambda_min, H^-1 = W diag(1/v) W^T, dp = -H^-1 g).  Only the lowest
// eigenvalue and one solve are needed, so the full spectrum is never formed:
//   1. H = Q T Q^T, blocked Householder tridiagonalisation (panels of 16 columns, updates of the
//      trailing matrix delayed to one fp64-MFMA rank-32 update per panel; the per-column products
//      A v stream the trailing matrix from L2 with 16-byte loads);
//   2. lambda_min(T) by multisection of the Sturm count (1024 shifts per round, 5-6 rounds);
//   3. nu = mu + rho |lambda_min| if lambda_min < lambda_min_threshold (augmented Hessian), else 0;
//   4. dp = -Q (T + nu I)^-1 Q^T g: compact-WY application of Q per panel, tridiagonal solve with
//      partial pivoting.
// The whole problem lives in one workgroup (LDS: the panel's V and W, vectors; global: a working
// copy of H and the stored reflectors), so a batch of G problems is G independent workgroups.
#include "common.h"
#include <math.h>

#ifdef OOVQE_NEWTON_TIMING
// tools/newton_probe.hip: cycles per phase, accumulated by thread 0 of workgroup 0
__device__ long long g_newton_cycles[16];
#define NT_MARK(k)                                                                     \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                     \
            const long long now__ = clock64();                                         \
            g_newton_cycles[k] += now__ - t_mark;                                      \
            t_mark = now__;                                                            \
        }                                                                              \
    } while (0)
#else
#define NT_MARK(k) do {} while (0)
#endif

namespace {

constexpr int NT = 1024;        // threads per workgroup (thread i <-> row i of the problem)
constexpr int NW = NT / 64;     // waves
constexpr int NB = 16;          // panel width
constexpr int NEWTON_NMAX = 480;

// Wave-wide sum on the VALU data-parallel-primitive path (row_shr 1/2/4/8, row_bcast 15/31: the
// total lands in lane 63 and is broadcast through an SGPR) -- a dozen VALU instructions instead of
// six dependent LDS-crossbar shuffles per value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_take<0x111, 0xf>(v);
    v += dpp_take<0x112, 0xf>(v);
    v += dpp_take<0x114, 0xf>(v);
    v += dpp_take<0x118, 0xf>(v);
    v += dpp_take<0x142, 0xa>(v);
    v += dpp_take<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Second stage of a workgroup reduction: the per-wave partial results (a, b) of NW = 16 waves sit in
// LDS as [wave][2]; every lane fetches the pair of wave (lane & 15) with one 16-byte read and the
// 16 values are summed inside each row of 16 lanes on the DPP path (lane 15 of a row holds the
// total) -- instead of 16 dependent LDS reads per thread.
__device__ __forceinline__ void row16_total2(const double* pairs, int lane, double& a, double& b)
{
    const d2 x = *reinterpret_cast<const d2*>(pairs + 2 * (lane & 15));
    a = x.x;
    b = x.y;
    a += dpp_take<0x111, 0xf>(a); b += dpp_take<0x111, 0xf>(b);
    a += dpp_take<0x112, 0xf>(a); b += dpp_take<0x112, 0xf>(b);
    a += dpp_take<0x114, 0xf>(a); b += dpp_take<0x114, 0xf>(b);
    a += dpp_take<0x118, 0xf>(a); b += dpp_take<0x118, 0xf>(b);
    a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 15),
                         __builtin_amdgcn_readlane(__double2loint(a), 15));
    b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b), 15),
                         __builtin_amdgcn_readlane(__double2loint(b), 15));
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), i.e. it would wait for the prefetched row of the next column and for the
// reflector's store to the workspace at every one of the four barriers of a column.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 1 / x to fp64 rounding-level accuracy without the division macro's scale / fixup steps (the
// operands here are far from the exponent limits): hardware estimate + two Newton steps
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Layout {     // dynamic LDS, in doubles
    int npv;        // pitch of Vc / Wc rows, >= n + 16, multiple of 16
    int npb;        // length of the v vector, >= n + 128
    int pcap;       // parts the rows of the symv are dealt to (partial sums [pcap][npv])
    int Vc, Wc, vb, pb, x1, x2, Tl, Gm, red, dd, ee, bb, aux, total;
};

__host__ __device__ inline Layout make_layout(int n)
{
    Layout L;
    L.npv = ((n + 16 + 15) / 16) * 16;
    L.npb = ((n + 128 + 15) / 16) * 16;
    {   // threads per part at the first column: pairs of columns, rounded to whole waves
        const int tpp0 = ((((n + 1) / 2) + 63) / 64) * 64;
        const int p0 = NT / (tpp0 > 0 ? tpp0 : 64);
        L.pcap = p0 < 1 ? 1 : (p0 > 8 ? 8 : p0);
    }
    int o = 0;
    L.Vc = o; o += NB * L.npv;
    L.Wc = o; o += NB * L.npv;
    L.vb = o; o += L.npb;
    L.pb = o; o += L.pcap * L.npv;
    L.x1 = o; o += NB;
    L.x2 = o; o += NB;
    L.Tl = o; o += NB * NB;
    L.Gm = o; o += NB * NB;
    L.red = o; o += 4 * 2 * NW + 2;
    // tridiagonal phase (after the panels): aliases of the V / W panel storage
    L.dd = L.Vc;                 // diagonal               [n]
    L.ee = L.Vc + L.npv;         // off-diagonal           [n]
    L.bb = L.Vc + 2 * L.npv;     // right-hand side        [n]
    L.aux = L.Vc + 3 * L.npv;    // dgtsv work: dl, du, dw [3 n]
    L.total = o;
    return L;
}

// sum of (a, b) over the workgroup; `red` holds two alternating buffers of 2*NW doubles
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red, int& parity, int lane, int wave)
{
    a = wave_sum(a);
    b = wave_sum(b);
    double* r = red + parity * 2 * NW;
    if (lane == 0) { r[2 * wave] = a; r[2 * wave + 1] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sa += r[2 * w]; sb += r[2 * w + 1]; }
    a = sa; b = sb;
    parity ^= 1;
}

__device__ __forceinline__ void block_min2(double& a, double& b, double* red, int& parity, int lane, int wave)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = fmin(a, __shfl_xor(a, o, 64));
        b = fmin(b, __shfl_xor(b, o, 64));
    }
    double* r = red + parity * 2 * NW;
    if (lane == 0) { r[2 * wave] = a; r[2 * wave + 1] = b; }
    __syncthreads();
    double sa = r[0], sb = r[1];
#pragma unroll
    for (int w = 1; w < NW; ++w) { sa = fmin(sa, r[2 * w]); sb = fmin(sb, r[2 * w + 1]); }
    a = sa; b = sb;
    parity ^= 1;
}

// work layout per problem (doubles): Aw [n][lda] | Vst [npan][NB][npv] | Tst [npan][NB*NB]
__host__ __device__ inline size_t newton_work_per_problem(int n)
{
    const Layout L = make_layout(n);
    const int lda = (n + 1) & ~1;
    const int npan = n > 1 ? (n - 1 + NB - 1) / NB : 0;
    size_t w = (size_t)n * lda + (size_t)npan * NB * L.npv + (size_t)npan * NB * NB + 16;
    return (w + 1) & ~(size_t)1;
}

__global__ __launch_bounds__(NT)
void newton_direction_kernel(const double* __restrict__ H, const double* __restrict__ g, int n,
                             double lam_threshold, double mu, double rho, int aug,
                             double* __restrict__ work, double* __restrict__ dp,
                             double* __restrict__ lowest, double* __restrict__ shift_out)
{
    extern __shared__ double sm[];
    const Layout L = make_layout(n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lda = (n + 1) & ~1;
    const int npan = n > 1 ? (n - 1 + NB - 1) / NB : 0;
    const int npv = L.npv;
    double* Vc = sm + L.Vc;
    double* Wc = sm + L.Wc;
    double* vb = sm + L.vb;
    double* pb = sm + L.pb;
    double* x1 = sm + L.x1;
    double* x2 = sm + L.x2;
    double* Tl = sm + L.Tl;
    double* Gm = sm + L.Gm;      // [l][j], l < j: (V^T v_j)[l]; [j][j]: tau_j
    double* red = sm + L.red;
    int parity = 0;

    const double* Hb = H + (size_t)blockIdx.x * n * n;
    const double* gb = g + (size_t)blockIdx.x * n;
    double* Aw = work + (size_t)blockIdx.x * newton_work_per_problem(n);
    double* Vst = Aw + (size_t)n * lda;
    double* Tst = Vst + (size_t)npan * NB * npv;
    // d and e of the tridiagonal matrix are kept in registers of "their" thread until the panels
    // are done (thread c owns d[c] and e[c]) and then parked in LDS
    double my_d = 0.0, my_e = 0.0;

#ifdef OOVQE_NEWTON_TIMING
    long long t_mark = clock64();
#endif
    // working copy of H, rows padded to an even pitch (pad column = 0)
    for (int idx = tid; idx < n * lda; idx += NT) {
        const int r = idx / lda, c = idx - r * lda;
        Aw[idx] = c < n ? Hb[(size_t)r * n + c] : 0.0;
    }
    for (int idx = tid; idx < L.npb; idx += NT) vb[idx] = 0.0;
    __syncthreads();

    const int i = tid;          // this thread's row
    NT_MARK(0);
    double* redA = red;                 // [2 NW] first reduction of a column: (|x|^2, alpha)
    double* redB = red + 2 * NW;        // [2 NW + 1] second one: (p^T v, -) per wave, then p at row c + 1
    double* part = pb;                  // [pcap][npv] partial sums of A22 v
    for (int pi = 0; pi < npan; ++pi) {
        const int k0 = pi * NB;
        const int jb = (n - 1 - k0) < NB ? (n - 1 - k0) : NB;
        for (int idx = tid; idx < NB * npv; idx += NT) { Vc[idx] = 0.0; Wc[idx] = 0.0; }
        if (tid < NB * NB) { Tl[tid] = 0.0; Gm[tid] = 0.0; }
        // a short last panel: its unused reflector rows must read as zero in the Q application
        for (int idx = jb * npv + tid; idx < NB * npv; idx += NT) Vst[(size_t)pi * NB * npv + idx] = 0.0;
        // row k0 of the matrix (== column k0: the full square is kept symmetric), fresh after the
        // previous panel's update; the rows of the following columns are prefetched one column ahead
        double rownext = i < n ? Aw[(size_t)k0 * lda + i] : 0.0;
        // the panel's newest reflector at this thread's row (v, w) and its w at the next column's row:
        // its rank-2 term is applied from registers, so no barrier separates a column from the next
        double v_prev = 0.0, w_prev = 0.0, w_prev_c = 0.0;
        __syncthreads();
        for (int j = 0; j < jb; ++j) {
            const int c = k0 + j;
            // ---- column c with the panel's pending rank-2 updates applied
            double colv = 0.0;
            if (i >= c && i < n) {
                // older reflectors of the panel from LDS, four at a time (independent partial sums, so
                // the sixteen LDS reads of a group are in flight together)
                double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
                int l = 0;
                for (; l + 4 < j; l += 4) {
                    c0 += Vc[l * npv + i] * Wc[l * npv + c] + Wc[l * npv + i] * Vc[l * npv + c];
                    c1 += Vc[(l + 1) * npv + i] * Wc[(l + 1) * npv + c] + Wc[(l + 1) * npv + i] * Vc[(l + 1) * npv + c];
                    c2 += Vc[(l + 2) * npv + i] * Wc[(l + 2) * npv + c] + Wc[(l + 2) * npv + i] * Vc[(l + 2) * npv + c];
                    c3 += Vc[(l + 3) * npv + i] * Wc[(l + 3) * npv + c] + Wc[(l + 3) * npv + i] * Vc[(l + 3) * npv + c];
                }
                for (; l + 1 < j; ++l)
                    c0 += Vc[l * npv + i] * Wc[l * npv + c] + Wc[l * npv + i] * Vc[l * npv + c];
                colv = rownext - ((c0 + c1) + (c2 + c3));
                if (j > 0) colv -= v_prev * w_prev_c + w_prev;          // V[j-1][c] = 1
            }
            if (i == c) my_d = colv;
            {
                const double s2w = wave_sum((i > c + 1 && i < n) ? colv * colv : 0.0);
                const double alw = wave_sum((i == c + 1) ? colv : 0.0);
                if (lane == 0) { redA[2 * wave] = s2w; redA[2 * wave + 1] = alw; }
            }
            double rowpre = 0.0;
            if (j + 1 < jb && i < n) rowpre = Aw[(size_t)(c + 1) * lda + i];
            NT_MARK(1);
            lds_barrier();
            NT_MARK(2);
            double s2, al;
            row16_total2(redA, lane, s2, al);
            // ---- Householder reflector (every thread, redundantly): H_c = I - tau v v^T
            double beta, tau, scale;
            if (s2 == 0.0) { beta = al; tau = 0.0; scale = 0.0; }
            else {
                beta = -copysign(sqrt(al * al + s2), al);
                tau = (beta - al) / beta;
                scale = 1.0 / (al - beta);
            }
            if (i == c) my_e = beta;
            const double vi = (i == c + 1) ? 1.0 : ((i > c + 1 && i < n) ? colv * scale : 0.0);
            if (i < npv) { vb[i] = vi; Vc[j * npv + i] = vi; Vst[((size_t)pi * NB + j) * npv + i] = vi; }
            lds_barrier();
            NT_MARK(3);
            // ---- x1 = W^T v, x2 = V^T v over the panel's earlier columns (one wave per column)
            if (wave < j) {
                double a1 = 0.0, a2 = 0.0;
                for (int r = c + 1 + lane; r < n; r += 64) {
                    const double vv = vb[r];
                    a1 += Wc[wave * npv + r] * vv;
                    a2 += Vc[wave * npv + r] * vv;
                }
                a1 = wave_sum(a1);
                a2 = wave_sum(a2);
                if (lane == 0) { x1[wave] = a1; x2[wave] = a2; Gm[wave * NB + j] = a2; }
            }
            NT_MARK(4);
            // ---- p_raw = A22 v on the trailing matrix as of the panel's start, streamed from L2.
            // Column-parallel: a thread owns two adjacent columns and, A being symmetric, sums
            // A[r][col] v[r] down its share of the rows (coalesced 16-byte loads, v[r] a broadcast
            // LDS read, no cross-lane reduction); the rows are dealt to P groups of threads whose
            // partial sums meet in LDS.
            const int cs = (c + 1) & ~1;
            const int ncol2 = (n - cs + 1) >> 1;
            const int tpp = ((ncol2 + 63) >> 6) << 6;
            const int P = (NT / tpp) < L.pcap ? (NT / tpp) : L.pcap;
            {
                const int prt = tid / tpp, pos = tid - prt * tpp;
                if (prt < P && pos < ncol2) {
                    const int col = cs + 2 * pos;
                    const double* ap = Aw + col;
                    double ax = 0.0, ay = 0.0;
                    int r = c + 1 + prt;
                    for (; r + 5 * P < n; r += 6 * P) {
                        d2 a[6];
                        double vv[6];
#pragma unroll
                        for (int u = 0; u < 6; ++u) {
                            a[u] = *reinterpret_cast<const d2*>(ap + (size_t)(r + u * P) * lda);
                            vv[u] = vb[r + u * P];
                        }
#pragma unroll
                        for (int u = 0; u < 6; ++u) { ax += a[u].x * vv[u]; ay += a[u].y * vv[u]; }
                    }
                    for (; r < n; r += P) {
                        const d2 a = *reinterpret_cast<const d2*>(ap + (size_t)r * lda);
                        const double vv = vb[r];
                        ax += a.x * vv;
                        ay += a.y * vv;
                    }
                    *reinterpret_cast<d2*>(part + prt * npv + col) = d2{ax, ay};
                }
            }
            NT_MARK(5);
            lds_barrier();
            NT_MARK(6);
            // ---- p = tau (A22 v - V (W^T v) - W (V^T v)),  w = p - (tau/2)(p^T v) v
            double pi_ = 0.0;
            if (i > c && i < n) {
                double sacc = 0.0;
                for (int k = 0; k < P; ++k) sacc += part[k * npv + i];
                double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
                int l = 0;
                for (; l + 3 < j; l += 4) {
                    c0 += Vc[l * npv + i] * x1[l] + Wc[l * npv + i] * x2[l];
                    c1 += Vc[(l + 1) * npv + i] * x1[l + 1] + Wc[(l + 1) * npv + i] * x2[l + 1];
                    c2 += Vc[(l + 2) * npv + i] * x1[l + 2] + Wc[(l + 2) * npv + i] * x2[l + 2];
                    c3 += Vc[(l + 3) * npv + i] * x1[l + 3] + Wc[(l + 3) * npv + i] * x2[l + 3];
                }
                for (; l < j; ++l) c0 += Vc[l * npv + i] * x1[l] + Wc[l * npv + i] * x2[l];
                pi_ = tau * (sacc - ((c0 + c1) + (c2 + c3)));
            }
            {
                const double pvw = wave_sum(pi_ * vi);
                if (lane == 0) { redB[2 * wave] = pvw; redB[2 * wave + 1] = 0.0; }
                if (i == c + 1) redB[2 * NW] = pi_;
            }
            lds_barrier();
            NT_MARK(7);
            double pv, pv_unused;
            row16_total2(redB, lane, pv, pv_unused);
            const double hp = 0.5 * tau * pv;
            const double wi = pi_ - hp * vi;
            if (i < npv) Wc[j * npv + i] = wi;
            if (tid == 0) Gm[j * NB + j] = tau;       // diagonal of the Gram table holds tau_j
            v_prev = vi;
            w_prev = wi;
            w_prev_c = redB[2 * NW] - hp;      // w at row c + 1, where v = 1
            rownext = rowpre;
            NT_MARK(8);
        }
        __syncthreads();                       // the last column's W is complete
        // compact WY factor of the panel, Q_panel = I - V T V^T with T upper triangular:
        // T[j][j] = tau_j, T[:j, j] = -tau_j T[:j, :j] (V^T v_j)[:j]; row l of T is built by thread l
        // (column after column, the rows are independent of each other given the earlier columns)
        if (tid < NB) {
            for (int jj = 0; jj < jb; ++jj) {
                const double tj = Gm[jj * NB + jj];
                double val = 0.0;
                if (tid == jj) val = tj;
                else if (tid < jj) {
                    double sacc = 0.0;
                    for (int mm = tid; mm < jj; ++mm) sacc += Tl[tid * NB + mm] * Gm[mm * NB + jj];
                    val = -tj * sacc;
                }
                Tl[tid * NB + jj] = val;
            }
        }
        // ---- trailing matrix -= V W^T + W V^T (rows and columns from r0 on), 16x16 tiles on the
        // fp64 matrix cores: lane supplies A[m = lane&15][k = lane>>4], B[k][n = lane&15]; a wave
        // keeps two tiles in flight (the tile's old values are loaded before the products run)
        {
            const int r0 = k0 + jb;
            const int mt = (n - r0 + 15) / 16;
            const int ntile = mt * mt;
            const int lq = lane >> 4, lr = lane & 15;
            for (int tile = wave; tile < ntile; tile += 2 * NW) {
                int rowb[2], colb[2];
                double old[2][4];
                d4 acc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tt = tile + h * NW < ntile ? tile + h * NW : tile;
                    const int ti = tt / mt, tj = tt - ti * mt;
                    rowb[h] = r0 + 16 * ti;
                    colb[h] = r0 + 16 * tj;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = rowb[h] + lq + 4 * e, col = colb[h] + lr;
                        old[h][e] = (row < n && col < n) ? Aw[(size_t)row * lda + col] : 0.0;
                    }
                    acc[h] = d4{0.0, 0.0, 0.0, 0.0};
                }
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const int k = lq + 4 * sx;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        acc[h] = mfma_f64(Vc[k * npv + rowb[h] + lr], Wc[k * npv + colb[h] + lr], acc[h]);
                        acc[h] = mfma_f64(Wc[k * npv + rowb[h] + lr], Vc[k * npv + colb[h] + lr], acc[h]);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && tile + NW >= ntile) break;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = rowb[h] + lq + 4 * e, col = colb[h] + lr;
                        if (row < n && col < n) Aw[(size_t)row * lda + col] = old[h][e] - acc[h][e];
                    }
                }
            }
        }
        __syncthreads();
        if (tid < NB * NB) Tst[(size_t)pi * NB * NB + tid] = Tl[tid];
        __syncthreads();
        NT_MARK(9);
    }
    if (n >= 1 && i == n - 1) my_d = Aw[(size_t)(n - 1) * lda + (n - 1)];

    // ---------------- tridiagonal phase ----------------
    double* dd = sm + L.dd;
    double* ee = sm + L.ee;
    double* bb = sm + L.bb;
    double* aux = sm + L.aux;
    double* e2 = sm + L.Vc + 6 * npv;          // squared off-diagonal elements [n]
    if (i < n) {
        const double ei = (i < n - 1) ? my_e : 0.0;
        dd[i] = my_d; ee[i] = ei; e2[i] = ei * ei; bb[i] = -gb[i];
    }
    __syncthreads();
    // Gershgorin lower bound and the smallest diagonal element bracket lambda_min
    double lo, hi, emax2;
    {
        double glo = INFINITY, dmin = INFINITY, nem = 0.0, dum = INFINITY;
        if (i < n) {
            const double el = i > 0 ? fabs(ee[i - 1]) : 0.0, er = i < n - 1 ? fabs(ee[i]) : 0.0;
            glo = dd[i] - el - er;
            dmin = dd[i];
            nem = -(er * er);
        }
        block_min2(glo, dmin, red, parity, lane, wave);
        block_min2(nem, dum, red, parity, lane, wave);
        lo = glo; hi = dmin; emax2 = -nem;
    }
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax2);
    {
        const double span = fmax(fabs(lo), fabs(hi));
        lo -= 2.0 * 2.220446049250313e-16 * span + 2.0 * pivmin;     // count(lo) == 0 stays true under rounding
        hi += 2.0 * 2.220446049250313e-16 * span + 2.0 * pivmin;     // count(hi) >= 1
    }
    for (int round = 0; round < 9; ++round) {
        const double width = hi - lo;
        if (!(width > 4.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + 4.0 * pivmin)) break;
        const double x = lo + width * ((double)(tid + 1) / (double)(NT + 1));
        // Sturm sequence of T - x: is there an eigenvalue below x?  (no early exit, reciprocal
        // instead of a division: the loop is one dependent chain of n steps per thread)
        double q = dd[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        bool below = q < 0.0;
#pragma unroll 4
        for (int k = 1; k < n; ++k) {
            q = (dd[k] - x) - e2[k - 1] * fast_rcp(q);
            if (fabs(q) < pivmin) q = -pivmin;
            below |= q < 0.0;
        }
        double first = below ? (double)tid : (double)NT, dum = 0.0;
        block_min2(first, dum, red, parity, lane, wave);
        const int t0 = (int)first;            // first shift with an eigenvalue below it (NT: none, then hi stays)
        const double nlo = t0 > 0 ? lo + width * ((double)t0 / (double)(NT + 1)) : lo;
        const double nhi = t0 < NT ? lo + width * ((double)(t0 + 1) / (double)(NT + 1)) : hi;
        lo = nlo; hi = nhi;
    }
    NT_MARK(10);
    const double lam = 0.5 * (lo + hi);
    const double nu = (aug && lam < lam_threshold) ? mu + rho * fabs(lam) : 0.0;

    // ---- b <- Q^T b, panel by panel: b -= V T^T (V^T b)
    for (int pi = 0; pi < npan; ++pi) {
        const double* Vp = Vst + (size_t)pi * NB * npv;
        {
            double a = 0.0;
            for (int r = lane; r < n; r += 64) a += Vp[(size_t)wave * npv + r] * bb[r];
            a = wave_sum(a);
            if (lane == 0) x1[wave] = a;
        }
        __syncthreads();
        if (tid < NB) {
            double s = 0.0;
            for (int mm = 0; mm < NB; ++mm) s += Tst[(size_t)pi * NB * NB + mm * NB + tid] * x1[mm];
            x2[tid] = s;
        }
        __syncthreads();
        if (i < n) {
            double s = bb[i];
#pragma unroll
            for (int l = 0; l < NB; ++l) s -= Vp[(size_t)l * npv + i] * x2[l];
            bb[i] = s;
        }
        __syncthreads();
    }
    NT_MARK(11);
    // ---- (T + nu I) y = b: Gaussian elimination with partial pivoting on the tridiagonal matrix
    if (tid == 0) {
        // Row k enters step k as (dk, uk | bk) = (diagonal, super-diagonal | right-hand side); the
        // next row's sub-diagonal, diagonal and super-diagonal are still the original ones.  The
        // running row lives in registers; finished rows go to LDS as (1/diagonal, U, U2 | B).
        double* U = aux;             // super-diagonal of the eliminated rows
        double* U2 = aux + n;        // second super-diagonal (fill-in of a row interchange)
        double dk = dd[0] + nu, uk = n > 1 ? ee[0] : 0.0, bk = bb[0];
        for (int k = 0; k + 1 < n; ++k) {
            const double sub = ee[k], dn = dd[k + 1] + nu, un = (k + 2 < n) ? ee[k + 1] : 0.0, bn = bb[k + 1];
            if (fabs(dk) >= fabs(sub)) {
                const double rd = 1.0 / dk;
                const double fact = sub * rd;
                dd[k] = rd; U[k] = uk; U2[k] = 0.0; bb[k] = bk;
                dk = dn - fact * uk; uk = un; bk = bn - fact * bk;
            } else {
                const double rs = 1.0 / sub;
                const double fact = dk * rs;
                dd[k] = rs; U[k] = dn; U2[k] = un; bb[k] = bn;
                dk = uk - fact * dn; uk = -fact * un; bk = bk - fact * bn;
            }
        }
        double x1v = bk / dk, x2v = 0.0;          // back substitution, two newest unknowns in registers
        bb[n - 1] = x1v;
        for (int k = n - 2; k >= 0; --k) {
            const double xk = (bb[k] - U[k] * x1v - U2[k] * x2v) * dd[k];
            bb[k] = xk;
            x2v = x1v;
            x1v = xk;
        }
    }
    __syncthreads();
    NT_MARK(12);
    // ---- y <- Q y, panels in reverse: y -= V T (V^T y)
    for (int pi = npan - 1; pi >= 0; --pi) {
        const double* Vp = Vst + (size_t)pi * NB * npv;
        {
            double a = 0.0;
            for (int r = lane; r < n; r += 64) a += Vp[(size_t)wave * npv + r] * bb[r];
            a = wave_sum(a);
            if (lane == 0) x1[wave] = a;
        }
        __syncthreads();
        if (tid < NB) {
            double s = 0.0;
            for (int mm = 0; mm < NB; ++mm) s += Tst[(size_t)pi * NB * NB + tid * NB + mm] * x1[mm];
            x2[tid] = s;
        }
        __syncthreads();
        if (i < n) {
            double s = bb[i];
#pragma unroll
            for (int l = 0; l < NB; ++l) s -= Vp[(size_t)l * npv + i] * x2[l];
            bb[i] = s;
        }
        __syncthreads();
    }
    NT_MARK(13);
    if (i < n) dp[(size_t)blockIdx.x * n + i] = bb[i];
    if (tid == 0) {
        lowest[blockIdx.x] = lam;
        if (shift_out) shift_out[blockIdx.x] = nu;
    }
}

}  // namespace

extern "C" int oovqe_newton_direction_max_n(void) { return NEWTON_NMAX; }

extern "C" int64_t oovqe_newton_direction_work_size(int n, int batch)
{
    if (n < 1 || n > NEWTON_NMAX || batch < 1) return 0;
    return (int64_t)newton_work_per_problem(n) * batch;
}

extern "C" int oovqe_newton_direction(const double* hessian, const double* gradient, int n, int batch,
                                      double lambda_min, double mu, double rho, int aug, double* work,
                                      double* dp, double* lowest_eigenvalue, double* shift,
                                      oovqe_stream_t stream)
{
    OOVQE_REQUIRE(hessian && gradient && work && dp && lowest_eigenvalue, "oovqe_newton_direction: null pointer");
    OOVQE_REQUIRE(n >= 1 && n <= NEWTON_NMAX, "oovqe_newton_direction: n = %d outside 1..%d", n, NEWTON_NMAX);
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "oovqe_newton_direction: batch = %d", batch);
    const Layout L = make_layout(n);
    const size_t lds = (size_t)L.total * sizeof(double);
    OOVQE_REQUIRE(lds <= 160 * 1024, "oovqe_newton_direction: %zu bytes of LDS needed", lds);
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)newton_direction_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    hipLaunchKernelGGL(newton_direction_kernel, dim3(batch), dim3(NT), lds, (hipStream_t)stream, hessian,
                       gradient, n, lambda_min, mu, rho, aug, work, dp, lowest_eigenvalue, shift);
    OOVQE_CHECK_LAUNCH("oovqe_newton_direction");
    return 0;
}
