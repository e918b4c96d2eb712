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


// =====================================================================================================
// Round 3: the same direction from SEVERAL workgroups per problem, in two stages.
//
//   stage 1  H = Q1 Bnd Q1^T, Bnd symmetric with half-bandwidth BW = 8: blocked two-sided Householder
//            reduction, panels of BW columns.  Everything that touches the n x n trailing matrix is a
//            matrix-matrix product on the fp64 matrix cores, spread over the W workgroups of the problem
//            by 16-row tiles (tile t belongs to workgroup t mod W and stays there):
//              P  (the tile owner of the panel's rows)  QR of the (m x BW) panel in LDS, one wave per
//                 column -> V, T (compact WY), the band entries of these BW columns;
//              X  (all)  X0 = A22 V for the own row tiles (reads only rows this workgroup wrote itself);
//              U  (all)  W = X0 T - 1/2 V (T^T V^T X0 T), own rows of A22 -= V W^T + W V^T.
//            Two hand-offs per panel (V | T from the panel owner to all, X0 rows from all to all).  The
//            exchanged doubles validate themselves: their buffers start as an all-ones NaN pattern no
//            computation produces, every location is written once (write-through, `sc1`) and read with
//            L1-bypassing `sc1` loads until no lane sees the pattern -- no flags, no fences, no ordering.
//   stage 2  (workgroup 0 of the problem) lambda_min(Bnd) by multisection on the test "Bnd - x I is
//            positive definite" = band LDL^T without pivoting, ONE SHIFT PER LANE (the 9 x 9 active
//            window of the factorisation lives in 45 registers, ring-indexed so that nothing moves; 256
//            shifts per round, 8 bits per round); nu as before; (Bnd + nu I) y = Q1^T (-g) by the same
//            LDL^T (positive definite by construction when aug != 0); dp = Q1 y.
// Per column nothing is left that synchronises more than one workgroup; per panel two hand-offs of
// ~1.5 us.  n <= NEWTON2_NMAX (LDS of the panel operands).
// =====================================================================================================
#ifdef OOVQE_NEWTON_TIMING
// tools/newton2_probe.hip: cycles per phase of the two-stage kernels, thread 0 of problem 0's workgroup 0
__device__ long long g_newton2_cycles[16];
#define N2_MARK(k)                                                                     \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                     \
            const long long now__ = clock64();                                         \
            g_newton2_cycles[k] += now__ - t_mark;                                     \
            t_mark = now__;                                                            \
        }                                                                              \
    } while (0)
#else
#define N2_MARK(k) do {} while (0)
#endif

constexpr int BW = 8;                 // band half-width = panel width
constexpr int RW = BW + 1;            // rows / columns of the LDL^T window
constexpr int NEWTON2_NMAX = 672;
constexpr int N2_SPIN_LIMIT = 1 << 17;
constexpr int N2_SHIFTS = 256;        // shifts per multisection round and workgroup (4 waves, one per SIMD)
constexpr int N2_MS_ROUNDS = 16;      // at most this many multisection rounds
constexpr int N2_MS_WGS = 128;        // at most this many workgroups share the shifts of a round

typedef unsigned n2_v4u __attribute__((ext_vector_type(4)));
typedef unsigned n2_v2u __attribute__((ext_vector_type(2)));
constexpr int N2_SC1 = 16;            // gfx950 buffer aux bit: sc1 (bypass L1 on loads, write through on stores)

struct N2Global {                     // workspace of a call: [Aw of every problem | exchange block of every problem | status words]
    int npv, npan, ntile;
    size_t aw_size;                   // doubles per problem in the first block (the working copy, [npv][npv])
    size_t Vst, Tst, X0, Band, bvec, msx, ex_size;   // offsets inside a problem's exchange block, and its size
};

__host__ __device__ inline N2Global n2_global(int n)
{
    N2Global L;
    L.ntile = (n + 15) / 16;
    L.npv = 16 * L.ntile;
    L.npan = n >= 2 ? (n - 2) / BW : 0;
    L.aw_size = (size_t)L.npv * L.npv;
    size_t o = 0;
    L.Vst = o; o += (size_t)L.npan * BW * L.npv;
    L.Tst = o; o += (size_t)L.npan * BW * BW;
    L.X0 = o; o += (size_t)L.npan * L.npv * BW;
    L.Band = o; o += (size_t)L.npv * RW;
    L.bvec = o; o += (size_t)L.npv;              // Q1^T (-g), from stage 1 to the solve kernel
    L.msx = o; o += (size_t)N2_MS_ROUNDS * N2_MS_WGS;   // multisection: first failing shift of every workgroup, per round
    L.ex_size = (o + 1) & ~(size_t)1;
    return L;
}
// doubles of workspace for `batch` problems (2 status doubles per problem at the end)
__host__ __device__ inline size_t n2_work_total(const N2Global& L, int batch)
{
    return (size_t)batch * (L.aw_size + L.ex_size + 2);
}

constexpr int RQ_SMALL = 6, RQ_LARGE = (NEWTON2_NMAX - BW + 63) / 64;   // panel-column elements per lane of the owning wave

struct N2Lds {                        // dynamic LDS of stage 1, offsets in doubles
    int ldp;                          // pitch of the VW / Xc rows: npv + 4 (rows 8 banks apart)
    int VW, Xc, part, Tm, Gm, Ym, Zm, S0, TQ, tau, x1, x2, bb, total;
};

__host__ __device__ inline N2Lds n2_lds(int n)
{
    const int npv = 16 * ((n + 15) / 16);
    N2Lds L;
    L.ldp = npv + 4;
    int o = 0;
    L.VW = o; o += 16 * L.ldp;        // rows 0..7 = V, 8..15 = W
    L.Xc = o; o += 8 * L.ldp;         // X0 [j][r]
    L.part = o; o += 16 * 128;        // K-split partial tiles of X0 / of V^T X0; the panel column during the QR (2 x 64 RQ_LARGE <= 2048)
    L.Tm = o; o += 64;
    L.Gm = o; o += 64;
    L.Ym = o; o += 64;
    L.Zm = o; o += 64;
    L.S0 = o; o += 64;
    L.TQ = o; o += 128;               // [T ; -1/2 Y] (16 x 8)
    L.tau = o; o += 16;
    L.x1 = o; o += 16;
    L.x2 = o; o += 16;
    L.bb = o; o += npv + 16;
    L.total = o;
    return L;
}

__device__ __forceinline__ bool n2_skip(const double* __restrict__ info, int prob, int which)
{
    if (!info || which == 0) return false;
    const bool fast = info[prob] == 1.0;
    return which == 1 ? fast : !fast;
}

__device__ __forceinline__ bool n2_is_sent(double x)
{
    return (unsigned long long)__double_as_longlong(x) == ~0ull;
}

__device__ __forceinline__ d2 n2_ld2(__amdgpu_buffer_rsrc_t r, size_t elem)
{
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(elem * 8), 0, N2_SC1));
}
__device__ __forceinline__ double n2_ld1(__amdgpu_buffer_rsrc_t r, size_t elem)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(elem * 8), 0, N2_SC1));
}
__device__ __forceinline__ void n2_st2(__amdgpu_buffer_rsrc_t r, size_t elem, d2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(n2_v4u, v), r, (unsigned)(elem * 8), 0, N2_SC1);
}
__device__ __forceinline__ void n2_st1(__amdgpu_buffer_rsrc_t r, size_t elem, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(n2_v2u, v), r, (unsigned)(elem * 8), 0, N2_SC1);
}

__device__ __forceinline__ double n2_lane(double x, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src),
                            __builtin_amdgcn_readlane(__double2loint(x), src));
}

// index of the ring pair (a, b), 0 <= a, b < RW, in the 45-element triangle
__device__ __forceinline__ constexpr int n2_tri(int a, int b)
{
    return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a;
}

// Band LDL^T of (Bnd - sigma I) without pivoting, run by ONE LANE: rb[r][t] = Bnd[r][r - BW + t] (row r of
// the band, t = BW the diagonal; rows n .. n + 2 RW are zero).  Returns false when a pivot is not positive
// (the matrix is not positive definite: an eigenvalue <= sigma exists).  SOLVE: also carries the right-hand
// side through the elimination and leaves the solution of (Bnd - sigma I) y = rhs in rhs (LDS); only used
// on positive definite matrices.  The window of the factorisation is ring-indexed (row r lives in ring slot
// r mod 9) and the k loop unrolled by 9: every register index is static, no entry ever moves.  The loop
// runs over whole rounds of 9 (the rows behind n are zero rows: their pivots are ignored); the band row
// that enters the window is fetched a step ahead, and the pivot of the next step -- and its reciprocal --
// is formed before the other 35 updates of a step, which then run in the shadow of that dependent chain.
template <bool SOLVE>
__device__ __forceinline__ bool n2_band_ldlt(const double* __restrict__ rb, int n, double sigma, double pivmin,
                             double* __restrict__ rhs, double* __restrict__ Lst, double* __restrict__ dst,
                             double* __restrict__ zst)
{
    double E[RW * (RW + 1) / 2];
    double cw[RW];
#pragma unroll
    for (int a = 0; a < RW; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double v = rb[a * RW + BW - (a - b)];      // A[a][b], a >= b
            if (a == b) v -= sigma;
            E[n2_tri(a, b)] = v;
        }
        if (SOLVE) cw[a] = rhs[a];                     // (rhs is zero behind n)
    }
    bool ok = true;
    double rn[RW];                                     // band row k + RW, fetched one step ahead
#pragma unroll
    for (int t = 0; t < RW; ++t) rn[t] = rb[RW * RW + t];
    double rinv = fast_rcp(E[n2_tri(0, 0)]);
    for (int k0 = 0; k0 < n; k0 += RW) {
#pragma unroll
        for (int s = 0; s < RW; ++s) {
            const int k = k0 + s;
            const int s1 = (s + 1) % RW;
            double rnn[RW];
            const double* rnp = rb + (size_t)(k + RW + 1) * RW;
#pragma unroll
            for (int t = 0; t < RW; ++t) rnn[t] = rnp[t];
            ok = ok && (E[n2_tri(s, s)] > pivmin || k >= n);
            double col[RW], l[RW];
#pragma unroll
            for (int i = 1; i < RW; ++i) col[i] = E[n2_tri((s + i) % RW, s)];
            // the next pivot first, its reciprocal in flight while the rest of the window is updated
            l[1] = col[1] * rinv;
            E[n2_tri(s1, s1)] -= l[1] * col[1];
            const double rinv_next = fast_rcp(E[n2_tri(s1, s1)]);
#pragma unroll
            for (int i = 2; i < RW; ++i) {
                l[i] = col[i] * rinv;
                E[n2_tri((s + i) % RW, s1)] -= l[i] * col[1];
            }
#pragma unroll
            for (int i = 2; i < RW; ++i)
#pragma unroll
                for (int j = 2; j <= i; ++j)
                    E[n2_tri((s + i) % RW, (s + j) % RW)] -= l[i] * col[j];
            if (SOLVE) {
                const double zk = cw[s];
#pragma unroll
                for (int i = 1; i < RW; ++i) {
                    cw[(s + i) % RW] -= l[i] * zk;
                    Lst[k * BW + i - 1] = l[i];
                }
                dst[k] = rinv;
                zst[k] = zk;
                cw[s] = rhs[k + RW];
            }
            // ring slot s now holds row k + RW
#pragma unroll
            for (int t = 0; t < BW; ++t) E[n2_tri(s, (s + 1 + t) % RW)] = rn[t];
            E[n2_tri(s, s)] = rn[BW] - sigma;
#pragma unroll
            for (int t = 0; t < RW; ++t) rn[t] = rnn[t];
            rinv = rinv_next;
        }
    }
    if (SOLVE) {
        // L^T y = D^-1 z from the last row up; y overwrites rhs (rows >= n stay 0: their L entries are 0)
        for (int k = n - 1; k >= 0; --k) {
            double acc = zst[k] * dst[k];
#pragma unroll
            for (int i = 1; i < RW; ++i) acc -= Lst[k * BW + i - 1] * rhs[k + i];
            rhs[k] = acc;
        }
    }
    return ok;
}

// RQMAX: panel-column elements a lane holds during the QR (n - 8 <= 64 RQMAX)
template <int RQMAX>
__global__ __launch_bounds__(NT)
void newton_band_kernel(const double* __restrict__ H, const double* __restrict__ g, int n,
                        double* __restrict__ work, int W, int batch, const double* __restrict__ info, int which)
{
    extern __shared__ double sm[];
    // info[prob] == 1: the Cholesky fast path (newton_chol.hip) already delivered this problem's direction.
    // which = 1: only the problems it did not serve; which = 2: only those it did (their lowest eigenvalue).
    // Every workgroup of a problem takes the same decision, so nobody is left waiting for a peer.
    if (n2_skip(info, blockIdx.x % batch, which)) return;
    const N2Global GL = n2_global(n);
    const N2Lds L = n2_lds(n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int prob = blockIdx.x % batch, slot = blockIdx.x / batch;
    const int npv = GL.npv, lda = GL.npv, npan = GL.npan, ntile = GL.ntile, ldp = L.ldp;
    double* Aw = work + (size_t)prob * GL.aw_size;
    double* ex = work + (size_t)batch * GL.aw_size + (size_t)prob * GL.ex_size;
    // ra: the working copy (own rows are re-read past the L1); rs: the exchange block
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Aw, 0, (int)(GL.aw_size * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ex, 0, (int)(GL.ex_size * 8), 0x00020000);
    const double* Hb = H + (size_t)prob * n * n;
    double* VW = sm + L.VW;
    double* Xc = sm + L.Xc;
    double* part = sm + L.part;
    double* Tm = sm + L.Tm;
    double* Gm = sm + L.Gm;
    double* Ym = sm + L.Ym;
    double* Zm = sm + L.Zm;
    double* S0 = sm + L.S0;
    double* TQ = sm + L.TQ;
    double* taus = sm + L.tau;
    double* x1 = sm + L.x1;
    double* x2 = sm + L.x2;
    double* bb = sm + L.bb;
    bool dead = false;                // a hand-off timed out (never on a healthy run): give up loudly
    // status word of the problem: all ones (the memset) = healthy, 1 = a hand-off timed out
    int* status = reinterpret_cast<int*>(work + (size_t)batch * (GL.aw_size + GL.ex_size) + 2 * (size_t)prob);
#ifdef OOVQE_NEWTON_TIMING
    long long t_mark = clock64();
#endif

    // ---- working copy of the own row tiles, rows and columns padded with zeros to whole tiles
    for (int t = slot; t < ntile; t += W)
        for (int idx = tid; idx < 16 * npv; idx += NT) {
            const int r = 16 * t + idx / npv, c = idx % npv;
            Aw[(size_t)r * lda + c] = (r < n && c < n) ? Hb[(size_t)r * n + c] : 0.0;
        }
    // (no stale bits of LDS may ever meet a zero operand as a NaN)
    for (int idx = tid; idx < L.total; idx += NT) sm[idx] = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (slot == 0) {                  // b = -g, carried through the panels' Q^T by workgroup 0
        const double* gb = g + (size_t)prob * n;
        for (int i2 = tid; i2 < n; i2 += NT) bb[i2] = -gb[i2];
    }
    N2_MARK(0);

    for (int p = 0; p < npan && !dead; ++p) {
        const int k = p * BW, r0 = k + BW, m = n - r0;
        const int leader = (k >> 4) % W;
        const size_t vst_p = GL.Vst + (size_t)p * BW * npv, tst_p = GL.Tst + (size_t)p * BW * BW;
        const size_t x0_p = GL.X0 + (size_t)p * npv * BW;
        // =============== P: QR of the panel, by the owner of rows k .. k+7 ===============
        // Column c of the panel = row k + c of the (symmetric) working copy, columns r0 .. n-1: wave c keeps
        // it in registers, element i in lane i % 64.  A step j costs ONE barrier: wave j drops its column
        // into LDS, and then every wave forms the reflector for itself from that copy (norm, beta, tau,
        // scale: redundant and cheap) next to its own dot product with it -- nothing waits for wave j.
        if (slot == leader) {
            for (int idx = tid; idx < BW * ldp; idx += NT) VW[idx] = 0.0;
            if (tid < 64) { Tm[tid] = 0.0; Gm[tid] = 0.0; }
            if (tid < 16) taus[tid] = 0.0;
            // y: the wave's panel column; once its own step is done, its reflector (for the T factor)
            double y[RQMAX];
            double rcol = 0.0;                                // lanes i <= c: R[i][c] of the wave's column c
            const int jb = (m - 1) < BW ? (m - 1) : BW;
            if (wave < BW) {
#pragma unroll
                for (int q = 0; q < RQMAX; ++q) {
                    const int i2 = lane + 64 * q;
                    // (rows behind the matrix are zero rows of the working copy: in range for q < RQMAX)
                    y[q] = i2 < m ? n2_ld1(ra, (size_t)(k + wave) * lda + r0 + (i2 < m ? i2 : 0)) : 0.0;
                }
                // band: the part of column k + wave inside the diagonal block
                if (lane < BW - wave)
                    n2_st1(rs, GL.Band + (size_t)(k + wave) * RW + lane,
                           n2_ld1(ra, (size_t)(k + wave) * lda + k + wave + lane));
            }
            N2_MARK(1);
            for (int j = 0; j < jb; ++j) {
                double* colb = part + (j & 1) * (64 * RQMAX);
                if (wave == j) {
#pragma unroll
                    for (int q = 0; q < RQMAX; ++q) colb[lane + 64 * q] = y[q];
                }
                lds_barrier();
                if (wave < BW) {
                    // (only the first 64 elements of a column can sit at or above the step's row j < 8: the
                    // others take part in everything without a select)
                    double x[RQMAX];
                    x[0] = colb[lane];
                    const double xb0 = lane > j ? x[0] : 0.0;                    // elements i > j
                    double s2 = xb0 * xb0, dc = xb0 * y[0];
#pragma unroll
                    for (int q = 1; q < RQMAX; ++q) {
                        x[q] = colb[lane + 64 * q];
                        s2 = fma(x[q], x[q], s2);
                        dc = fma(x[q], y[q], dc);
                    }
                    s2 = wave_sum(s2);
                    dc = wave_sum(dc);
                    const double al = n2_lane(x[0], j), yj = n2_lane(y[0], j);
                    const bool nz = s2 != 0.0;
                    const double beta = nz ? -copysign(sqrt(fma(al, al, s2)), al) : al;
                    const double tau = nz ? (beta - al) * fast_rcp(beta) : 0.0;
                    const double scale = nz ? fast_rcp(al - beta) : 0.0;
                    const double v0 = lane > j ? x[0] * scale : (lane == j ? 1.0 : 0.0);
                    if (wave > j) {
                        const double f = tau * fma(scale, dc, yj), fs = f * scale;
                        y[0] = fma(-f, v0, y[0]);
#pragma unroll
                        for (int q = 1; q < RQMAX; ++q) y[q] = fma(-fs, x[q], y[q]);
                    } else if (wave == j) {
                        rcol = lane == j ? beta : y[0];
                        y[0] = v0;
                        VW[j * ldp + r0 + (lane < m ? lane : m)] = lane < m ? v0 : 0.0;
#pragma unroll
                        for (int q = 1; q < RQMAX; ++q) {
                            const int i2 = lane + 64 * q;
                            y[q] = x[q] * scale;
                            // (lanes behind the panel all drop a zero just behind the matrix: inside the row's pad)
                            VW[j * ldp + r0 + (i2 < m ? i2 : m)] = y[q];
                        }
                        if (lane == 0) taus[j] = tau;
                    } else {
                        // an earlier reflector (now in y): (V^T v_j)[wave] = v_wave[j] + scale * sum_{i > j} v_wave[i] x[i]
                        const double gv = fma(scale, dc, yj);
                        if (lane == 0) Gm[wave * BW + j] = gv;
                    }
                }
            }
            N2_MARK(2);
            // band: the R part (row r0 + i of column k + wave, i <= wave)
            if (wave < BW && wave >= jb) rcol = y[0];
            if (wave < BW && lane <= wave && lane < m)
                n2_st1(rs, GL.Band + (size_t)(k + wave) * RW + BW + lane - wave, rcol);
            __syncthreads();
            // compact WY factor: row l of T by thread l (upper triangular), the row and the products V^T v it needs
            // in registers (28 independent LDS reads, then arithmetic only; the sums run over mm = l .. jj - 1 as
            // before: the elements left of the diagonal are zeros)
            if (tid < BW) {
                double trow[BW], gc[BW][BW], tj[BW];
#pragma unroll
                for (int jj = 0; jj < BW; ++jj) {
                    tj[jj] = taus[jj];
                    trow[jj] = 0.0;
#pragma unroll
                    for (int mm = 0; mm < jj; ++mm) gc[mm][jj] = Gm[mm * BW + jj];
                }
#pragma unroll
                for (int jj = 0; jj < BW; ++jj) {
                    double sacc = 0.0;
#pragma unroll
                    for (int mm = 0; mm < jj; ++mm) sacc += trow[mm] * gc[mm][jj];
                    const double val = tid == jj ? tj[jj] : (tid < jj ? -tj[jj] * sacc : 0.0);
                    trow[jj] = jj < jb ? val : 0.0;
                }
#pragma unroll
                for (int jj = 0; jj < BW; ++jj) Tm[tid * BW + jj] = trow[jj];
            }
            __syncthreads();
            for (int idx = tid; idx < BW * (npv / 2); idx += NT) {
                const int j = idx / (npv / 2), c2 = idx - j * (npv / 2);
                n2_st2(rs, vst_p + (size_t)j * npv + 2 * c2, *reinterpret_cast<const d2*>(VW + j * ldp + 2 * c2));
            }
            if (tid < 32) n2_st2(rs, tst_p + 2 * (size_t)tid, *reinterpret_cast<const d2*>(Tm + 2 * tid));
            N2_MARK(3);
        } else {
            // =============== X (1): V | T from the panel owner ===============
            int spins = 0;
            while (true) {
                int bad = 0;
                for (int idx = tid; idx < BW * (npv / 2); idx += NT) {
                    const int j = idx / (npv / 2), c2 = idx - j * (npv / 2);
                    const d2 v = n2_ld2(rs, vst_p + (size_t)j * npv + 2 * c2);
                    bad |= (int)n2_is_sent(v.x) | (int)n2_is_sent(v.y);
                    *reinterpret_cast<d2*>(VW + j * ldp + 2 * c2) = v;
                }
                if (tid < 32) {
                    const d2 v = n2_ld2(rs, tst_p + 2 * (size_t)tid);
                    bad |= (int)n2_is_sent(v.x) | (int)n2_is_sent(v.y);
                    *reinterpret_cast<d2*>(Tm + 2 * tid) = v;
                }
                if (!__syncthreads_or(bad)) break;
                ++spins;
                int st = -1;
                if ((spins & 63) == 0) st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__syncthreads_or(st == 1 || spins > N2_SPIN_LIMIT)) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            N2_MARK(4);
        }
        // =============== X (2): X0 = A22 V on the own row tiles ===============
        const int cg0 = r0 >> 4;                 // first 16-column group / row tile that reaches into A22
        int q0 = 0;                              // own tiles t = slot + W q, q >= q0, reach into A22
        while (slot + W * q0 < cg0) ++q0;
        int n_own = 0;
        for (int t = slot + W * q0; t < ntile; t += W) ++n_own;
        {
            const int KS = n_own > 0 && n_own < NW ? NW / n_own : 1;     // K split over the waves
            for (int item = wave; item < n_own * KS; item += NW) {
                const int qi = item / KS, ks = item - qi * KS;
                const int t = slot + W * (q0 + qi);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const size_t arow = (size_t)(16 * t + lr) * lda + 4 * lq;
                for (int gq = cg0 + ks; gq < ntile; gq += KS) {
                    const d2 a01 = n2_ld2(ra, arow + 16 * gq);
                    const d2 a23 = n2_ld2(ra, arow + 16 * gq + 2);
                    const double* vb = VW + (lr & 7) * ldp + 16 * gq + 4 * lq;
                    acc = mfma_f64(a01.x, vb[0], acc);
                    acc = mfma_f64(a01.y, vb[1], acc);
                    acc = mfma_f64(a23.x, vb[2], acc);
                    acc = mfma_f64(a23.y, vb[3], acc);
                }
                if (KS == 1) {
                    if (lr < BW)
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2)
                            n2_st1(rs, x0_p + (size_t)(16 * t + lq + 4 * i2) * BW + lr, acc[i2]);
                } else if (lr < BW) {
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) part[item * 128 + (lq + 4 * i2) * BW + lr] = acc[i2];
                }
            }
            N2_MARK(5);
            if (KS > 1) {
                __syncthreads();
                for (int idx = tid; idx < n_own * 128; idx += NT) {
                    const int qi = idx >> 7, e = idx & 127;
                    double sacc = 0.0;
                    for (int ks = 0; ks < KS; ++ks) sacc += part[(qi * KS + ks) * 128 + e];
                    const int t = slot + W * (q0 + qi);
                    n2_st1(rs, x0_p + (size_t)(16 * t) * BW + e, sacc);
                }
            }
        }
        N2_MARK(6);
        // =============== U: all of X0, W, own rows of A22 -= V W^T + W V^T ===============
        {
            int spins = 0;
            while (true) {
                int bad = 0;
                for (int idx = tid; idx < (n - r0) * (BW / 2); idx += NT) {
                    const int r = r0 + idx / (BW / 2), j2 = (idx % (BW / 2)) * 2;
                    const d2 v = n2_ld2(rs, x0_p + (size_t)r * BW + j2);
                    bad |= (int)n2_is_sent(v.x) | (int)n2_is_sent(v.y);
                    Xc[j2 * ldp + r] = v.x;
                    Xc[(j2 + 1) * ldp + r] = v.y;
                }
                // the rows of the first tile above the panel carry no X0
                if (tid < BW * 16) {
                    const int r = 16 * cg0 + (tid & 15);
                    if (r < r0) Xc[(tid >> 4) * ldp + r] = 0.0;
                }
                if (!__syncthreads_or(bad)) break;
                ++spins;
                int st = -1;
                if ((spins & 63) == 0) st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__syncthreads_or(st == 1 || spins > N2_SPIN_LIMIT)) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
        }
        N2_MARK(7);
        {   // S0 = V^T X0 (8 x 8) on the matrix cores, the rows dealt to the waves four at a time
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* va = VW + (lr & 7) * ldp + lq;
            const double* xb = Xc + (lr & 7) * ldp + lq;
            for (int r = r0 + 4 * wave; r < n; r += 4 * NW) acc = mfma_f64(va[r], xb[r], acc);
            if (lr < BW) {
                part[wave * 64 + lq * BW + lr] = acc[0];            // rows lq, lq + 4 of the 8 x 8 block
                part[wave * 64 + (lq + 4) * BW + lr] = acc[1];
            }
        }
        __syncthreads();
        if (tid < 64) {
            double sacc = 0.0;
#pragma unroll
            for (int c = 0; c < NW; ++c) sacc += part[c * 64 + tid];
            S0[tid] = sacc;
        }
        __syncthreads();
        if (tid < 64) {                           // Z = S0 T
            const int a = tid >> 3, b = tid & 7;
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < BW; ++q) sacc += S0[a * BW + q] * Tm[q * BW + b];
            Zm[tid] = sacc;
        }
        __syncthreads();
        if (tid < 64) {                           // Y = T^T Z;  TQ = [T ; -1/2 Y]
            const int a = tid >> 3, b = tid & 7;
            double sacc = 0.0;
#pragma unroll
            for (int l = 0; l < BW; ++l) sacc += Tm[l * BW + a] * Zm[l * BW + b];
            Ym[tid] = sacc;
            TQ[tid] = Tm[tid];
            TQ[64 + tid] = -0.5 * sacc;
        }
        __syncthreads();
        // W = [X0 | V] [T ; -1/2 Y]: one 16 x 16 x 16 product per row tile (rows outside r0 .. n-1 come out 0)
        for (int t = cg0 + wave; t < ntile; t += NW) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const int kk = 4 * sx + lq;
                const double av = kk < BW ? Xc[kk * ldp + 16 * t + lr] : VW[(kk - BW) * ldp + 16 * t + lr];
                acc = mfma_f64(av, TQ[kk * BW + (lr & 7)], acc);
            }
            if (lr < BW)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) VW[(BW + lr) * ldp + 16 * t + lq + 4 * i2] = acc[i2];
        }
        __syncthreads();
        N2_MARK(8);
        {
            const int ng = ntile - cg0;
            for (int item = wave; item < n_own * ng; item += NW) {
                const int qi = item / ng, ct = cg0 + item - qi * ng;
                const int t = slot + W * (q0 + qi);
                double old[4];
                const size_t base = (size_t)(16 * t + lq) * lda + 16 * ct + lr;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) old[i2] = n2_ld1(ra, base + (size_t)(4 * i2) * lda);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const int kk = 4 * sx + lq;
                    acc = mfma_f64(VW[kk * ldp + 16 * t + lr], VW[((kk + BW) & 15) * ldp + 16 * ct + lr], acc);
                }
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    Aw[(size_t)(16 * t + lq + 4 * i2) * lda + 16 * ct + lr] = old[i2] - acc[i2];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        N2_MARK(9);
        if (slot == 0) {
            // b <- Q_p^T b = b - V T^T (V^T b) (V | T of this panel are still in LDS), while the next panel's owner
            // factorises: this workgroup would only be waiting for it
            if (wave < BW) {
                double a = 0.0;
                for (int r = r0 + lane; r < n; r += 64) a += VW[wave * ldp + r] * bb[r];
                a = wave_sum(a);
                if (lane == 0) x1[wave] = a;
            }
            __syncthreads();
            if (tid < BW) {
                double sacc = 0.0;
                for (int mm = 0; mm < BW; ++mm) sacc += Tm[mm * BW + tid] * x1[mm];
                x2[tid] = sacc;
            }
            __syncthreads();
            for (int r = r0 + tid; r < n; r += NT) {
                double sacc = bb[r];
#pragma unroll
                for (int l = 0; l < BW; ++l) sacc -= VW[l * ldp + r] * x2[l];
                bb[r] = sacc;
            }
            __syncthreads();          // (the next panel refills VW)
        }
        N2_MARK(10);
    }
    // ---- band: the dense remainder behind the last panel (rows >= kend), by the owners of those rows
    if (!dead) {
        const int kend = npan * BW;
        for (int idx = tid; idx < (n - kend) * RW; idx += NT) {
            const int r = kend + idx / RW, c = r - (idx % RW);
            if (c >= kend && ((r >> 4) % W) == slot)
                n2_st1(rs, GL.Band + (size_t)c * RW + (r - c), n2_ld1(ra, (size_t)r * lda + c));
        }
        // b = Q1^T (-g) for the solve kernel
        if (slot == 0) {
            __syncthreads();
            for (int i2 = tid; i2 < n; i2 += NT) ex[GL.bvec + i2] = bb[i2];
        }
    }
    // a hand-off that timed out leaves its mark for the solve kernel (which then answers with NaNs)
    if (dead && tid == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// =============== stage 2: one workgroup of 256 threads per problem, launched behind stage 1 ===============
// (a kernel of its own: one wave per SIMD, so the one-shift-per-lane LDL^T has the whole register file --
// inside the 1024-thread stage-1 kernel it would be capped at 128 VGPRs and spill its window)
constexpr int NT2 = 256;
constexpr int NW2 = NT2 / 64;

struct N2Lds2 { int rb, Lst, dst, zst, Vp, Tm, x1, x2, red, bb, total; };

__host__ __device__ inline N2Lds2 n2_lds2(int n)
{
    const int npv = 16 * ((n + 15) / 16);
    N2Lds2 L;
    int o = 0;
    L.rb = o; o += (n + 2 * RW + 2) * RW;
    o = (o + 1) & ~1;
    L.Lst = o; o += (n + RW) * BW;
    L.dst = o; o += n + RW;
    L.zst = o; o += n + RW;
    o = (o + 1) & ~1;
    L.Vp = o; o += BW * npv;
    L.Tm = o; o += 64;
    L.x1 = o; o += 16;
    L.x2 = o; o += 16;
    L.red = o; o += 4 * 2 * NW + 2;
    L.bb = o; o += npv + 2 * RW + 16;
    L.total = o;
    return L;
}

__device__ __forceinline__ void block_min2_w4(double& a, double& b, double* red, int& parity, int lane, int wave)
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
    for (int w = 1; w < NW2; ++w) { sa = fmin(sa, r[2 * w]); sb = fmin(sb, r[2 * w + 1]); }
    a = sa; b = sb;
    parity ^= 1;
}

// BIG (n > NEWTON2_NMAX, the host-orchestrated stage 1 below): the band rows and the factor of the solve do
// not fit LDS; they live in `scratch` (global, (n + 2 RW + 2) RW + (n + RW)(BW + 2) doubles per problem), V is
// applied straight from memory, and `work` is that problem's exchange block (status = `*big_status`).
template <bool BIG>
__global__ __launch_bounds__(NT2)
void newton_band_solve_kernel(int n, double lam_threshold, double mu, double rho, int aug,
                              const double* __restrict__ work, double* __restrict__ dp,
                              double* __restrict__ lowest, double* __restrict__ shift_out,
                              double* __restrict__ scratch, int W2, int batch, double* __restrict__ info, int which)
{
    extern __shared__ double sm[];
    if (!BIG && n2_skip(info, blockIdx.x % batch, which)) return;
    const N2Global GL = n2_global(n);
    const N2Lds2 L = n2_lds2(BIG ? 16 : n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // W2 workgroups per problem share the shifts of every multisection round (small batches leave most of
    // the chip idle here); workgroup 0 of the problem goes on to the solve
    const int prob = blockIdx.x % batch, slot2 = blockIdx.x / batch;
    // the fast path served this problem: only its lowest eigenvalue is still wanted
    const bool lowest_only = !BIG && info && info[prob] == 1.0;
    const int npv = GL.npv, npan = GL.npan;
    const double* wk = BIG ? work : work + (size_t)batch * GL.aw_size + (size_t)prob * GL.ex_size;     // the exchange block
    double* rb = BIG ? scratch : sm + L.rb;
    double* Lst = BIG ? rb + (size_t)(n + 2 * RW + 2) * RW : sm + L.Lst;
    double* dst = BIG ? Lst + (size_t)(n + RW) * BW : sm + L.dst;
    double* zst = BIG ? dst + (n + RW) : sm + L.zst;
    double* Vp = sm + L.Vp;
    double* Tm = sm + L.Tm;
    double* x1 = sm + L.x1;
    double* x2 = sm + L.x2;
    double* red = sm + L.red;
    double* bb = BIG ? sm + L.total : sm + L.bb;            // (BIG: behind the small arrays, npv + 2 RW + 16 doubles)
    int parity = 0;
    const int dead = BIG ? 0 : *reinterpret_cast<const int*>(work + (size_t)batch * (GL.aw_size + GL.ex_size) + 2 * (size_t)prob) == 1;
#ifdef OOVQE_NEWTON_TIMING
    long long t_mark = clock64();
#endif
    for (int idx = tid; idx < (n + 2 * RW + 2) * RW; idx += NT2) rb[idx] = 0.0;
    for (int i2 = tid; i2 < npv + 2 * RW + 16; i2 += NT2) bb[i2] = i2 < n ? wk[GL.bvec + i2] : 0.0;
    __syncthreads();
    int bad = dead;
    {
        const double* Band = wk + GL.Band;
        for (int idx = tid; idx < n * RW; idx += NT2) {
            const int c = idx / RW, d = idx - c * RW;
            if (c + d < n) {
                const double v = Band[idx];
                bad |= (int)n2_is_sent(v);          // (an entry stage 1 never wrote: cannot happen on a healthy run)
                rb[(c + d) * RW + BW - d] = v;
            }
        }
    }
    if (__syncthreads_or(bad)) {
        if (slot2 != 0) return;
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        if (!lowest_only)
            for (int i2 = tid; i2 < n; i2 += NT2) dp[(size_t)prob * n + i2] = qnan;
        if (tid == 0) {
            lowest[prob] = qnan;
            if (shift_out && !lowest_only) shift_out[prob] = qnan;
            // a hand-off of stage 1 timed out.  The eigenvalue-only route (lowest_only; it may run on a side stream
            // BESIDE the route of the other problems) never writes info[]: the kernels of both routes decide what
            // to skip from it, so it has to stay what the fast path left.  Its report is the NaN in lowest[].
            if (info && !lowest_only) info[prob] = -1.0;
        }
        return;
    }
    // Gershgorin lower bound and the smallest diagonal element bracket lambda_min
    double lo, hi, amax;
    {
        double glo = INFINITY, dmin = INFINITY, nam = 0.0, dum = INFINITY;
        for (int r = tid; r < n; r += NT2) {
            const double dg = rb[r * RW + BW];
            double off = 0.0;
#pragma unroll
            for (int t = 0; t < BW; ++t) off += fabs(rb[r * RW + t]);
#pragma unroll
            for (int d = 1; d <= BW; ++d) off += fabs(rb[(r + d) * RW + BW - d]);
            glo = fmin(glo, dg - off);
            dmin = fmin(dmin, dg);
            nam = fmin(nam, -(off + fabs(dg)));
        }
        block_min2_w4(glo, dmin, red, parity, lane, wave);
        block_min2_w4(nam, dum, red, parity, lane, wave);
        lo = glo; hi = dmin; amax = -nam;            // amax = the infinity norm of the band matrix
    }
    N2_MARK(11);
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, amax * amax);
    {
        const double span = fmax(fabs(lo), fabs(hi));
        lo -= 2.0 * 2.220446049250313e-16 * span + 2.0 * pivmin;
        hi += 2.0 * 2.220446049250313e-16 * span + 2.0 * pivmin;
    }
    // bracket down to a rounding error of the matrix norm (what eigh delivers for an eigenvalue): NS = 256 W2
    // shifts per round, workgroup w tests the shifts 256 w .. 256 w + 255 and drops the index of its first failing
    // one into this round's slot of the exchange block (the same self-validating hand-off as in stage 1)
    const int NS = N2_SHIFTS * W2;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(wk), 0, (int)(GL.ex_size * 8), 0x00020000);
    bool lost = false;
    for (int round = 0; round < N2_MS_ROUNDS; ++round) {
        const double width = hi - lo;
        if (!(width > 2.220446049250313e-16 * amax + 4.0 * pivmin)) break;
        const int tg = slot2 * N2_SHIFTS + tid;
        double first = (double)NS, dum = 0.0;
        {
            const double x = lo + width * ((double)(tg + 1) / (double)(NS + 1));
            const bool pd = n2_band_ldlt<false>(rb, n, x, pivmin, nullptr, nullptr, nullptr, nullptr);
            if (!pd) first = (double)tg;
        }
        block_min2_w4(first, dum, red, parity, lane, wave);
        if (W2 > 1) {
            if (tid == 0) n2_st1(rx, GL.msx + (size_t)round * N2_MS_WGS + slot2, first);
            int spins = 0;
            while (true) {
                double v = (double)NS;
                if (tid < W2) v = n2_ld1(rx, GL.msx + (size_t)round * N2_MS_WGS + tid);
                const int bad2 = n2_is_sent(v);
                dum = 0.0;
                double vv = bad2 ? (double)NS : v;
                block_min2_w4(vv, dum, red, parity, lane, wave);
                if (!__syncthreads_or(bad2)) { first = vv; break; }
                if (++spins > N2_SPIN_LIMIT) { lost = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lost) break;
        }
        const int t0 = (int)first;            // first shift that is not below every eigenvalue
        const double nlo = t0 > 0 ? lo + width * ((double)t0 / (double)(NS + 1)) : lo;
        const double nhi = t0 < NS ? lo + width * ((double)(t0 + 1) / (double)(NS + 1)) : hi;
        lo = nlo; hi = nhi;
    }
    if (slot2 != 0) return;
    if (lost) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        if (!lowest_only)
            for (int i2 = tid; i2 < n; i2 += NT2) dp[(size_t)prob * n + i2] = qnan;
        if (tid == 0) {
            lowest[prob] = qnan;
            if (shift_out && !lowest_only) shift_out[prob] = qnan;
            if (info && !lowest_only) info[prob] = -1.0;
        }
        return;
    }
    const double lam = 0.5 * (lo + hi);
    const double nu = (aug && lam < lam_threshold) ? mu + rho * fabs(lam) : 0.0;
    N2_MARK(12);
    if (!(lam == lam) || !(amax <= 1.79e308)) {
        // a NaN or an Inf in the Hessian: nothing to compute, and nothing to hand on silently
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        if (!lowest_only)
            for (int i2 = tid; i2 < n; i2 += NT2) dp[(size_t)prob * n + i2] = qnan;
        if (tid == 0) {
            lowest[prob] = qnan;
            if (shift_out && !lowest_only) shift_out[prob] = qnan;
            if (info && !lowest_only) info[prob] = -3.0;
        }
        return;
    }
    if (lowest_only) {
        if (tid == 0) lowest[prob] = lam;
        return;
    }
    // without the level shift (aug == 0) the band LDL^T below has no pivoting: it is only valid on a positive
    // definite matrix.  An indefinite Hessian that is to be inverted as it stands is refused loudly (the
    // one-workgroup kernel with its pivoted tridiagonal solve serves n <= 480; Python falls back to eigh beyond)
    if (nu == 0.0 && !(lam > 0.0)) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        for (int i2 = tid; i2 < n; i2 += NT2) dp[(size_t)prob * n + i2] = qnan;
        if (tid == 0) {
            lowest[prob] = lam;
            if (shift_out) shift_out[prob] = 0.0;
            if (info) info[prob] = -2.0;
        }
        return;
    }
    // ---- (Bnd + nu I) y = b = Q1^T (-g) (carried through the panels by stage 1) by the same LDL^T, one lane
    if (tid == 0) n2_band_ldlt<true>(rb, n, -nu, 0.0, bb, Lst, dst, zst);
    __syncthreads();
    N2_MARK(14);
    // ---- dp = Q1 y: panels in reverse, y -= V T (V^T y); the next panel's V | T are fetched into registers
    // while the current one is applied
    constexpr int VREG = (BW * (16 * ((NEWTON2_NMAX + 15) / 16)) / 2 + NT2 - 1) / NT2;
    d2 vnext[VREG], tnext = {0.0, 0.0};
    auto fetch = [&](int p) {
        const double* vsrc = wk + GL.Vst + (size_t)p * BW * npv;
#pragma unroll
        for (int u = 0; u < VREG; ++u) {
            const int idx = tid + u * NT2;
            if (idx < BW * (npv / 2)) vnext[u] = *reinterpret_cast<const d2*>(vsrc + 2 * (size_t)idx);
        }
        if (tid < 32) tnext = *reinterpret_cast<const d2*>(wk + GL.Tst + (size_t)p * BW * BW + 2 * tid);
    };
    if (BIG) {
        for (int p = npan - 1; p >= 0; --p) {
            const double* vsrc = wk + GL.Vst + (size_t)p * BW * npv;
            const double* tsrc = wk + GL.Tst + (size_t)p * BW * BW;
            for (int col = wave; col < BW; col += NW2) {
                double a = 0.0;
                for (int r = lane; r < n; r += 64) a += vsrc[(size_t)col * npv + r] * bb[r];
                a = wave_sum(a);
                if (lane == 0) x1[col] = a;
            }
            __syncthreads();
            if (tid < BW) {
                double sacc = 0.0;
                for (int mm = 0; mm < BW; ++mm) sacc += tsrc[tid * BW + mm] * x1[mm];
                x2[tid] = sacc;
            }
            __syncthreads();
            for (int r = tid; r < n; r += NT2) {
                double sacc = bb[r];
#pragma unroll
                for (int l = 0; l < BW; ++l) sacc -= vsrc[(size_t)l * npv + r] * x2[l];
                bb[r] = sacc;
            }
            __syncthreads();
        }
    }
    if (!BIG && npan > 0) fetch(npan - 1);
    for (int p = BIG ? -1 : npan - 1; p >= 0; --p) {
#pragma unroll
        for (int u = 0; u < VREG; ++u) {
            const int idx = tid + u * NT2;
            if (idx < BW * (npv / 2)) *reinterpret_cast<d2*>(Vp + 2 * idx) = vnext[u];
        }
        if (tid < 32) *reinterpret_cast<d2*>(Tm + 2 * tid) = tnext;
        __syncthreads();
        if (p > 0) fetch(p - 1);
        for (int col = wave; col < BW; col += NW2) {
            double a = 0.0;
            for (int r = lane; r < n; r += 64) a += Vp[col * npv + r] * bb[r];
            a = wave_sum(a);
            if (lane == 0) x1[col] = a;
        }
        __syncthreads();
        if (tid < BW) {
            double sacc = 0.0;
            for (int mm = 0; mm < BW; ++mm) sacc += Tm[tid * BW + mm] * x1[mm];
            x2[tid] = sacc;
        }
        __syncthreads();
        for (int r = tid; r < n; r += NT2) {
            double sacc = bb[r];
#pragma unroll
            for (int l = 0; l < BW; ++l) sacc -= Vp[l * npv + r] * x2[l];
            bb[r] = sacc;
        }
        __syncthreads();
    }
    N2_MARK(15);
    for (int i2 = tid; i2 < n; i2 += NT2) dp[(size_t)prob * n + i2] = bb[i2];
    if (tid == 0) {
        lowest[prob] = lam;
        if (shift_out) shift_out[prob] = nu;
    }
}

// =====================================================================================================
// n > NEWTON2_NMAX (the panel operands no longer fit LDS; orbital spaces of N = 200: n ~ 4 700): the same
// band reduction, one launch per phase of a panel instead of in-kernel hand-offs (the phases are large
// enough to fill the chip by themselves, ~4 launches x n / 8 panels):
//   n2l_panel_kernel   QR of the panel (one workgroup; the columns in registers, 64 x RQ elements each)
//   n2l_x0_kernel      X0 = A22 V, a workgroup per 16-row tile, the four waves split K
//   n2l_w_kernel       S0, Y, W = X0 T - 1/2 V Y; b <- Q_p^T b
//   n2l_update_kernel  A22 -= V W^T + W V^T, a wave per 16 x 16 tile
// then the same solve kernel with its band rows and factor in global scratch.  n <= NEWTON3_NMAX.
// =====================================================================================================
constexpr int NEWTON3_NMAX = 5128;
constexpr int RQ_MID = 24;                                      // n - 8 <= 1536

struct N3Global {          // per problem, doubles: [Aw | Vst | Tst | Vt | X0 | Wt | Band | bvec | scratch]
    int npv, npan, ntile;
    size_t Aw, Vt, X0, Wt, scratch, small, total;
    N2Global ex;           // Vst / Tst / Band / bvec offsets are those of the exchange block (at `exoff`)
    size_t exoff;
};

__host__ __device__ inline N3Global n3_global(int n)
{
    N3Global L;
    L.ex = n2_global(n);
    L.npv = L.ex.npv; L.npan = L.ex.npan; L.ntile = L.ex.ntile;
    size_t o = 0;
    L.Aw = o; o += (size_t)L.npv * L.npv;
    L.exoff = o; o += L.ex.ex_size;
    L.Vt = o; o += (size_t)L.npv * BW;
    L.X0 = o; o += (size_t)L.npv * BW;
    L.Wt = o; o += (size_t)L.npv * BW;
    L.scratch = o; o += (size_t)(n + 2 * RW + 2) * RW + (size_t)(n + RW) * (BW + 2) + 16;
    L.small = o; o += (size_t)L.ntile * 72 + 160;      // per-tile partials of V^T X0 and V^T b | TQ (16 x 8) | x2
    L.total = (o + 1) & ~(size_t)1;
    return L;
}

__global__ void n2l_copy_kernel(const double* __restrict__ H, const double* __restrict__ g, int n, int npv,
                                double* __restrict__ Aw, double* __restrict__ bvec)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)npv * npv;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / npv), c = (int)(idx - (size_t)r * npv);
        Aw[idx] = (r < n && c < n) ? H[(size_t)r * n + c] : 0.0;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npv; i += gridDim.x * blockDim.x) bvec[i] = i < n ? -g[i] : 0.0;
}

// one workgroup of 8 waves: wave c owns panel column c (= row k + c of the working copy)
template <int RQMAX>
__global__ __launch_bounds__(512)
void n2l_panel_kernel(int n, int p, double* __restrict__ wk)
{
    extern __shared__ double sm[];
    const N3Global GL = n3_global(n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npv = GL.npv, lda = GL.npv;
    const int k = p * BW, r0 = k + BW, m = n - r0;
    const double* Aw = wk + GL.Aw;
    double* ex = wk + GL.exoff;
    double* Vst = ex + GL.ex.Vst + (size_t)p * BW * npv;
    double* Tst = ex + GL.ex.Tst + (size_t)p * BW * BW;
    double* Band = ex + GL.ex.Band;
    double* Vt = wk + GL.Vt;
    double* colbuf = sm;                          // [2][64 RQMAX]
    double* Tm = colbuf + 2 * 64 * RQMAX;         // [64]
    double* Gm = Tm + 64;
    double* taus = Gm + 64;
    if (tid < 64) { Tm[tid] = 0.0; Gm[tid] = 0.0; }
    if (tid < 16) taus[tid] = 0.0;
    for (int idx = tid; idx < BW * npv; idx += 512) Vst[idx] = 0.0;
    for (int idx = tid; idx < npv * BW; idx += 512) Vt[idx] = 0.0;
    double y[RQMAX];
    double rcol = 0.0;
    const int jb = (m - 1) < BW ? (m - 1) : BW;
#pragma unroll
    for (int q = 0; q < RQMAX; ++q) {
        const int i2 = lane + 64 * q;
        y[q] = i2 < m ? Aw[(size_t)(k + wave) * lda + r0 + i2] : 0.0;
    }
    if (lane < BW - wave) Band[(size_t)(k + wave) * RW + lane] = Aw[(size_t)(k + wave) * lda + k + wave + lane];
    __syncthreads();
    for (int j = 0; j < jb; ++j) {
        double* colb = colbuf + (j & 1) * (64 * RQMAX);
        if (wave == j) {
#pragma unroll
            for (int q = 0; q < RQMAX; ++q) colb[lane + 64 * q] = y[q];
        }
        lds_barrier();
        const double x0 = colb[lane];
        const double xb0 = lane > j ? x0 : 0.0;
        double s2 = xb0 * xb0, dc = xb0 * y[0];
#pragma unroll
        for (int q = 1; q < RQMAX; ++q) {
            const double xq = colb[lane + 64 * q];
            s2 = fma(xq, xq, s2);
            dc = fma(xq, y[q], dc);
        }
        s2 = wave_sum(s2);
        dc = wave_sum(dc);
        const double al = n2_lane(x0, j), yj = n2_lane(y[0], j);
        const bool nz = s2 != 0.0;
        const double beta = nz ? -copysign(sqrt(fma(al, al, s2)), al) : al;
        const double tau = nz ? (beta - al) * fast_rcp(beta) : 0.0;
        const double scale = nz ? fast_rcp(al - beta) : 0.0;
        const double v0 = lane > j ? x0 * scale : (lane == j ? 1.0 : 0.0);
        if (wave > j) {
            const double f = tau * fma(scale, dc, yj), fs = f * scale;
            y[0] = fma(-f, v0, y[0]);
#pragma unroll
            for (int q = 1; q < RQMAX; ++q) y[q] = fma(-fs, colb[lane + 64 * q], y[q]);
        } else if (wave == j) {
            rcol = lane == j ? beta : y[0];
            y[0] = v0;
            if (lane < m) { Vst[(size_t)j * npv + r0 + lane] = v0; Vt[(size_t)(r0 + lane) * BW + j] = v0; }
#pragma unroll
            for (int q = 1; q < RQMAX; ++q) {
                const int i2 = lane + 64 * q;
                y[q] = colb[i2] * scale;
                if (i2 < m) { Vst[(size_t)j * npv + r0 + i2] = y[q]; Vt[(size_t)(r0 + i2) * BW + j] = y[q]; }
            }
            if (lane == 0) taus[j] = tau;
        } else {
            const double gv = fma(scale, dc, yj);
            if (lane == 0) Gm[wave * BW + j] = gv;
        }
    }
    if (wave >= jb) rcol = y[0];
    if (lane <= wave && lane < m) Band[(size_t)(k + wave) * RW + BW + lane - wave] = rcol;
    __syncthreads();
    if (tid < BW) {
        for (int jj = 0; jj < jb; ++jj) {
            const double tj = taus[jj];
            double val = 0.0;
            if (tid == jj) val = tj;
            else if (tid < jj) {
                double sacc = 0.0;
                for (int mm = tid; mm < jj; ++mm) sacc += Tm[tid * BW + mm] * Gm[mm * BW + jj];
                val = -tj * sacc;
            }
            Tm[tid * BW + jj] = val;
        }
    }
    __syncthreads();
    if (tid < 64) Tst[tid] = Tm[tid];
}

// The same panel factorisation for long columns (m > 64 RQ_MID): the eight columns stay where they are -- rows
// k .. k+7 of the working copy, L2-resident -- and are streamed: 16 waves, two per column (even / odd blocks of
// 64 rows); per step one pass for the norm and the dot products and one for the update, two barriers.
__global__ __launch_bounds__(NT)
void n2l_panel_stream_kernel(int n, int p, double* __restrict__ wk)
{
    __shared__ double part[2][BW][2];         // [half][column][s2, dc]
    __shared__ double Tm[64], Gm[64], taus[16], Rm[64];
    const N3Global GL = n3_global(n);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = wave & 7, half = wave >> 3;
    const int npv = GL.npv, lda = GL.npv;
    const int k = p * BW, r0 = k + BW, m = n - r0;
    double* Aw = wk + GL.Aw;
    double* ex = wk + GL.exoff;
    double* Vst = ex + GL.ex.Vst + (size_t)p * BW * npv;
    double* Tst = ex + GL.ex.Tst + (size_t)p * BW * BW;
    double* Band = ex + GL.ex.Band;
    double* Vt = wk + GL.Vt;
    if (tid < 64) { Tm[tid] = 0.0; Gm[tid] = 0.0; Rm[tid] = 0.0; }
    if (tid < 16) taus[tid] = 0.0;
    for (int idx = tid; idx < BW * npv; idx += NT) Vst[idx] = 0.0;
    for (int idx = tid; idx < npv * BW; idx += NT) Vt[idx] = 0.0;
    if (half == 0 && lane < BW - col) Band[(size_t)(k + col) * RW + lane] = Aw[(size_t)(k + col) * lda + k + col + lane];
    __syncthreads();
    const int jb = (m - 1) < BW ? (m - 1) : BW;
    double* ycol = Aw + (size_t)(k + col) * lda + r0;       // this wave's column: y[i] = ycol[i]
    for (int j = 0; j < jb; ++j) {
        const double* x = Aw + (size_t)(k + j) * lda + r0;  // the pivot column (col < j: the stored reflector j' = col)
        const double* other = col < j ? Vst + (size_t)col * npv + r0 : ycol;
        // (element j of the own column is rewritten by the column's first wave behind the barrier below: read now)
        const double al = x[j], yj = other[j];
        double s2 = 0.0, dc = 0.0;
        for (int i0 = j + 1 + 64 * half; i0 < m; i0 += 512) {
            double xv[4], ov[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i2 = i0 + 128 * u + lane;
                xv[u] = i2 < m ? x[i2] : 0.0;
                ov[u] = i2 < m ? other[i2] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s2 = fma(xv[u], xv[u], s2); dc = fma(xv[u], ov[u], dc); }
        }
        s2 = wave_sum(s2);
        dc = wave_sum(dc);
        if (lane == 0) { part[half][col][0] = s2; part[half][col][1] = dc; }
        __syncthreads();
        s2 = part[0][col][0] + part[1][col][0];
        dc = part[0][col][1] + part[1][col][1];
        const bool nz = s2 != 0.0;
        const double beta = nz ? -copysign(sqrt(fma(al, al, s2)), al) : al;
        const double tau = nz ? (beta - al) * fast_rcp(beta) : 0.0;
        const double scale = nz ? fast_rcp(al - beta) : 0.0;
        if (col > j) {
            const double f = tau * fma(scale, dc, yj), fs = f * scale;
            for (int i0 = j + 1 + 64 * half; i0 < m; i0 += 512) {
                double xv[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i2 = i0 + 128 * u + lane;
                    xv[u] = i2 < m ? x[i2] : 0.0;
                    yv[u] = i2 < m ? ycol[i2] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i2 = i0 + 128 * u + lane;
                    if (i2 < m) ycol[i2] = fma(-fs, xv[u], yv[u]);
                }
            }
            if (half == 0 && lane == 0) { ycol[j] = yj - f; Rm[j * BW + col] = yj - f; }
        } else if (col == j) {
            for (int i2 = j + 1 + 64 * half + lane; i2 < m; i2 += 128) {
                const double v = x[i2] * scale;
                Vst[(size_t)j * npv + r0 + i2] = v;
                Vt[(size_t)(r0 + i2) * BW + j] = v;
            }
            if (half == 0 && lane == 0) {
                Vst[(size_t)j * npv + r0 + j] = 1.0;
                Vt[(size_t)(r0 + j) * BW + j] = 1.0;
                taus[j] = tau;
                Rm[j * BW + j] = beta;
            }
        } else if (half == 0 && lane == 0) {
            Gm[col * BW + j] = fma(scale, dc, yj);          // yj = v_col[j]
        }
        __syncthreads();
    }
    // rows of R that no step produced (a short last panel): the entries as they stand
    if (tid < 64) {
        const int i2 = tid >> 3, c = tid & 7;
        if (i2 >= jb && i2 <= c && i2 < m) Rm[i2 * BW + c] = Aw[(size_t)(k + c) * lda + r0 + i2];
    }
    __syncthreads();
    if (tid < 64) {
        const int i2 = tid >> 3, c = tid & 7;
        if (i2 <= c && i2 < m) Band[(size_t)(k + c) * RW + BW + i2 - c] = Rm[i2 * BW + c];
    }
    if (tid < BW) {
        for (int jj = 0; jj < jb; ++jj) {
            const double tj = taus[jj];
            double val = 0.0;
            if (tid == jj) val = tj;
            else if (tid < jj) {
                double sacc = 0.0;
                for (int mm = tid; mm < jj; ++mm) sacc += Tm[tid * BW + mm] * Gm[mm * BW + jj];
                val = -tj * sacc;
            }
            Tm[tid * BW + jj] = val;
        }
    }
    __syncthreads();
    if (tid < 64) Tst[tid] = Tm[tid];
}

// X0[r][0..7] = sum_c A[r][c] V[c][0..7] for the rows of tile blockIdx.x + cg0; 4 waves split the columns
__global__ __launch_bounds__(256)
void n2l_x0_kernel(int n, int p, double* __restrict__ wk)
{
    __shared__ double part[4][128];
    const N3Global GL = n3_global(n);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int lda = GL.npv, ntile = GL.ntile;
    const int r0 = p * BW + BW, cg0 = r0 >> 4;
    const int t = cg0 + blockIdx.x;
    const double* Aw = wk + GL.Aw;
    const double* Vt = wk + GL.Vt;
    double* X0 = wk + GL.X0;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* arow = Aw + (size_t)(16 * t + lr) * lda + 4 * lq;
    for (int gq = cg0 + wave; gq < ntile; gq += 4) {
        const d2 a01 = *reinterpret_cast<const d2*>(arow + 16 * gq);
        const d2 a23 = *reinterpret_cast<const d2*>(arow + 16 * gq + 2);
        const double* vb = Vt + (size_t)(16 * gq + 4 * lq) * BW + (lr & 7);
        acc = mfma_f64(a01.x, vb[0], acc);
        acc = mfma_f64(a01.y, vb[BW], acc);
        acc = mfma_f64(a23.x, vb[2 * BW], acc);
        acc = mfma_f64(a23.y, vb[3 * BW], acc);
    }
    if (lr < BW)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) part[wave][(lq + 4 * i2) * BW + lr] = acc[i2];
    __syncthreads();
    __shared__ double xs[128];
    if (tid < 128) {
        const int row = 16 * t + tid / BW;
        const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        xs[tid] = (row >= r0 && row < n) ? v : 0.0;
        X0[(size_t)row * BW + (tid & 7)] = xs[tid];
    }
    __syncthreads();
    // this tile's share of S0 = V^T X0 and of V^T b (summed over the tiles, in order, by n2l_w_kernel)
    double* sp = wk + GL.small + (size_t)blockIdx.x * 72;
    if (tid < 64) {
        const int a = tid >> 3, b = tid & 7;
        double sacc = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sacc += Vt[(size_t)(16 * t + rr) * BW + a] * xs[rr * BW + b];
        sp[tid] = sacc;
    } else if (tid < 72) {
        const int a = tid - 64;
        const double* bvec = wk + GL.exoff + GL.ex.bvec;
        double sacc = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sacc += Vt[(size_t)(16 * t + rr) * BW + a] * bvec[16 * t + rr];
        sp[tid] = sacc;
    }
}

// one workgroup: S0 (from the tiles' partial sums), Y = T^T S0 T, TQ = [T ; -1/2 Y] for the update kernel
// (which forms W = [X0 | V] TQ for its own rows); b <- Q_p^T b = b - V T^T (V^T b)
__global__ __launch_bounds__(256)
void n2l_w_kernel(int n, int p, double* __restrict__ wk)
{
    __shared__ double part[4][72];
    __shared__ double S0[64], Zm[64], Tm[64], x1[BW], x2[BW];
    const N3Global GL = n3_global(n);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = p * BW + BW, cg0 = r0 >> 4, ng = GL.ntile - cg0;
    const double* Vt = wk + GL.Vt;
    const double* sp = wk + GL.small;
    double* TQ = wk + GL.small + (size_t)GL.ntile * 72;
    double* bvec = wk + GL.exoff + GL.ex.bvec;
    const double* Tst = wk + GL.exoff + GL.ex.Tst + (size_t)p * BW * BW;
    if (tid < 64) Tm[tid] = Tst[tid];
    {
        double s0 = 0.0, s1 = 0.0;       // pair `lane` of S0; waves 0 .. 3 take every fourth tile; lanes < 8 also x1
        for (int t = wave; t < ng; t += 4) {
            s0 += sp[(size_t)t * 72 + lane];
            if (lane < BW) s1 += sp[(size_t)t * 72 + 64 + lane];
        }
        part[wave][lane] = s0;
        if (lane < BW) part[wave][64 + lane] = s1;
    }
    __syncthreads();
    if (tid < 64) S0[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    if (tid < BW) x1[tid] = (part[0][64 + tid] + part[1][64 + tid]) + (part[2][64 + tid] + part[3][64 + tid]);
    __syncthreads();
    if (tid < 64) {
        const int a = tid >> 3, b = tid & 7;
        double sacc = 0.0;
        for (int q = 0; q < BW; ++q) sacc += S0[a * BW + q] * Tm[q * BW + b];
        Zm[tid] = sacc;
    }
    if (tid >= 64 && tid < 64 + BW) {
        const int j = tid - 64;
        double sacc = 0.0;
        for (int mm = 0; mm < BW; ++mm) sacc += Tm[mm * BW + j] * x1[mm];
        x2[j] = sacc;
    }
    __syncthreads();
    if (tid < 64) {
        const int a = tid >> 3, b = tid & 7;
        double sacc = 0.0;
        for (int l = 0; l < BW; ++l) sacc += Tm[l * BW + a] * Zm[l * BW + b];
        TQ[tid] = Tm[tid];
        TQ[64 + tid] = -0.5 * sacc;
    }
    for (int r = r0 + tid; r < n; r += 256) {
        double sacc = bvec[r];
#pragma unroll
        for (int l = 0; l < BW; ++l) sacc -= Vt[(size_t)r * BW + l] * x2[l];
        bvec[r] = sacc;
    }
}

// a wave per 16 x 16 tile of the trailing matrix (tiles cg0 .. ntile-1 in both directions)
__global__ __launch_bounds__(512)
void n2l_update_kernel(int n, int p, double* __restrict__ wk)
{
    const N3Global GL = n3_global(n);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int lda = GL.npv, ntile = GL.ntile;
    const int r0 = p * BW + BW, cg0 = r0 >> 4, ng = ntile - cg0;
    const long item = (long)blockIdx.x * 8 + wave;
    if (item >= (long)ng * ng) return;
    const int t = cg0 + (int)(item / ng), ct = cg0 + (int)(item - (long)(item / ng) * ng);
    double* Aw = wk + GL.Aw;
    const double* Vt = wk + GL.Vt;
    const double* X0 = wk + GL.X0;
    const double* TQ = wk + GL.small + (size_t)GL.ntile * 72;       // [T ; -1/2 Y] (16 x 8)
    // W[r][j] = sum_l X0[r][l] T[l][j] - 1/2 V[r][l] Y[l][j] for this lane's rows of the two tiles, j = lq, lq + 4
    // (rows outside r0 .. n-1 carry zeros in X0 and V)
    double vrow[2][BW], wv[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const size_t r = (size_t)(16 * (h ? ct : t) + lr);
        double xr[BW];
#pragma unroll
        for (int l = 0; l < BW; ++l) { xr[l] = X0[r * BW + l]; vrow[h][l] = Vt[r * BW + l]; }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = lq + 4 * jj;
            double sacc = 0.0;
#pragma unroll
            for (int l = 0; l < BW; ++l) sacc += xr[l] * TQ[l * BW + j] + vrow[h][l] * TQ[(BW + l) * BW + j];
            wv[h][jj] = sacc;
        }
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
        // k = 4 sx + lq:  P = [V | W] of the row tile, Q = [W | V] of the column tile
        // (the V element of a lane depends on lq: fetched again rather than picked out of vrow[] by a lane-dependent index)
        const double pa = sx < 2 ? Vt[(size_t)(16 * t + lr) * BW + 4 * sx + lq] : wv[0][sx - 2];
        const double qb = sx < 2 ? wv[1][sx] : Vt[(size_t)(16 * ct + lr) * BW + 4 * (sx - 2) + lq];
        acc = mfma_f64(pa, qb, acc);
    }
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
        double* a = Aw + (size_t)(16 * t + lq + 4 * i2) * lda + 16 * ct + lr;
        *a -= acc[i2];
    }
}

__global__ void n2l_tail_kernel(int n, double* __restrict__ wk)
{
    const N3Global GL = n3_global(n);
    const int kend = GL.npan * BW, lda = GL.npv;
    const double* Aw = wk + GL.Aw;
    double* Band = wk + GL.exoff + GL.ex.Band;
    for (int idx = threadIdx.x; idx < (n - kend) * RW; idx += blockDim.x) {
        const int r = kend + idx / RW, c = r - (idx % RW);
        if (c >= kend) Band[(size_t)c * RW + (r - c)] = Aw[(size_t)r * lda + c];
    }
}

}  // namespace

static int n2_cu_count()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 1;
    }
    return cus;
}

int oovqe_newton_chol_launch(const double* hessian, const double* gradient, int n, int batch, double lambda_min,
                             double* work, double* dp, double* shift, double* info, hipStream_t st);
size_t oovqe_newton_chol_work(int n, int batch);
int oovqe_newton_chol_max_n(void);

// which kernel serves (n, aug): the two-stage multi-workgroup one whenever the level shift is on (then the
// system it solves is positive definite by construction); without the shift (aug == 0: the reference then
// inverts an indefinite Hessian as it stands) the one-workgroup kernel with its pivoted tridiagonal solve
static bool n2_use_band(int n, int aug)
{
    if (oovqe_opt(OOVQE_OPT_NEWTON_ONE_WG)) return false;
    if (n > NEWTON2_NMAX) return false;
    if (!aug && n <= NEWTON_NMAX) return false;
    return true;
}

// does the positive-definite fast path (newton_chol.hip) run in front of the band route?
static bool n2_use_chol(int n, int aug)
{
    return aug && n <= oovqe_newton_chol_max_n() && n2_use_band(n, aug) && !oovqe_opt(OOVQE_OPT_NEWTON_NO_CHOL);
}

extern "C" int oovqe_newton_direction_max_n(void) { return NEWTON3_NMAX; }
extern "C" int oovqe_newton_direction_pd_max_n(void) { return oovqe_newton_chol_max_n(); }
// 1 when oovqe_newton_direction(n, aug) runs the positive-definite fast path in front of the band route
extern "C" int oovqe_newton_direction_has_pd(int n, int aug) { return n >= 1 && n2_use_chol(n, aug) ? 1 : 0; }

// n > NEWTON2_NMAX: one problem after the other, ~4 launches per panel of 8 columns
static int n3_direction(const double* hessian, const double* gradient, int n, int batch, double lambda_min,
                        double mu, double rho, int aug, double* work, double* dp, double* lowest, double* shift,
                        double* info, hipStream_t st)
{
    const N3Global GL = n3_global(n);
    const size_t panel_lds = (size_t)(2 * 64 * RQ_MID + 64 + 64 + 16) * sizeof(double);
    const size_t solve_lds = (size_t)(n2_lds2(16).total + GL.npv + 2 * RW + 16) * sizeof(double);
    OOVQE_REQUIRE(panel_lds <= 159 * 1024 && solve_lds <= 159 * 1024, "oovqe_newton_direction: n = %d too large", n);
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)n2l_panel_kernel<RQ_MID>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)newton_band_solve_kernel<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    for (int b = 0; b < batch; ++b) {
        double* wk = work + (size_t)b * GL.total;
        const double* Hb = hessian + (size_t)b * n * n;
        hipLaunchKernelGGL(n2l_copy_kernel, dim3(1024), dim3(256), 0, st, Hb, gradient + (size_t)b * n, n, GL.npv,
                           wk + GL.Aw, wk + GL.exoff + GL.ex.bvec);
        for (int p = 0; p < GL.npan; ++p) {
            const int r0 = p * BW + BW, cg0 = r0 >> 4, ng = GL.ntile - cg0;
            // (the panel shrinks by 8 rows each time: the long-column kernel until the columns fit registers)
            if (n - r0 <= 64 * RQ_MID) hipLaunchKernelGGL(n2l_panel_kernel<RQ_MID>, dim3(1), dim3(512), panel_lds, st, n, p, wk);
            else hipLaunchKernelGGL(n2l_panel_stream_kernel, dim3(1), dim3(NT), 0, st, n, p, wk);
            hipLaunchKernelGGL(n2l_x0_kernel, dim3(ng), dim3(256), 0, st, n, p, wk);
            hipLaunchKernelGGL(n2l_w_kernel, dim3(1), dim3(256), 0, st, n, p, wk);
            hipLaunchKernelGGL(n2l_update_kernel, dim3((unsigned)(((long)ng * ng + 7) / 8)), dim3(512), 0, st, n, p, wk);
        }
        hipLaunchKernelGGL(n2l_tail_kernel, dim3(1), dim3(256), 0, st, n, wk);
        hipLaunchKernelGGL(newton_band_solve_kernel<true>, dim3(1), dim3(NT2), solve_lds, st, n, lambda_min, mu, rho,
                           aug, wk + GL.exoff, dp + (size_t)b * n, lowest + b, shift ? shift + b : nullptr,
                           wk + GL.scratch, 1, 1, info ? info + b : nullptr, 0);
        OOVQE_CHECK_LAUNCH("oovqe_newton_direction/large");
    }
    return 0;
}

// doubles of the band / one-workgroup route alone
static size_t n2_route_work(int n, int batch)
{
    if (n > NEWTON2_NMAX) return n3_global(n).total * (size_t)batch;
    const size_t a = n <= NEWTON_NMAX ? newton_work_per_problem(n) : 0;
    const size_t b = n2_work_total(n2_global(n), 1);
    return (a > b ? a : b) * (size_t)batch;
}

extern "C" int64_t oovqe_newton_direction_rest_work_size(int n, int batch)
{
    if (n < 1 || n > NEWTON3_NMAX || batch < 1) return 0;
    return (int64_t)n2_route_work(n, batch);
}

extern "C" int64_t oovqe_newton_direction_pd_work_size(int n, int batch)
{
    if (n < 1 || n > oovqe_newton_chol_max_n() || batch < 1) return 0;
    return (int64_t)oovqe_newton_chol_work(n, batch);
}

// [route work | Cholesky work | info] -- the one-call entry point runs the fast path in front of the band route
extern "C" int64_t oovqe_newton_direction_work_size(int n, int batch)
{
    if (n < 1 || n > NEWTON3_NMAX || batch < 1) return 0;
    size_t w = n2_route_work(n, batch);
    if (n <= oovqe_newton_chol_max_n()) w += oovqe_newton_chol_work(n, batch) + (size_t)batch + 2;
    return (int64_t)w;
}

// Workgroups per problem of a band-route launch: they wait for each other, so all batch * W of them must be
// resident together.  The bound is what the occupancy query says this kernel gets per CU with its LDS (one,
// at these sizes) times the CU count -- and only this process's launches are counted: when other work holds
// CUs a hand-off can time out (2^17 polls), which the solve kernel reports in info[] (-1) and
// oovqe_newton_direction_rest(..., max_wg = 1) then repeats without any inter-workgroup wait.
static int n2_resident_limit(const void* kernel, int threads, size_t lds)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (per_cu > 1) per_cu = 1;            // one workgroup per CU is what the hand-off latencies were measured at
    return per_cu * n2_cu_count();
}

static int n2_band_route(const double* hessian, const double* gradient, int n, int batch, double lambda_min, double mu,
                         double rho, int aug, double* work, double* dp, double* lowest_eigenvalue, double* shift,
                         double* info, int which, int max_wg, hipStream_t st)
{
    const N2Global GL = n2_global(n);
    const N2Lds L = n2_lds(n);
    const size_t lds = (size_t)L.total * sizeof(double);
    const size_t lds2 = (size_t)n2_lds2(n).total * sizeof(double);
    OOVQE_REQUIRE(lds <= 159 * 1024 && lds2 <= 159 * 1024, "oovqe_newton_direction: %zu bytes of LDS needed",
                  lds > lds2 ? lds : lds2);
    OOVQE_REQUIRE(GL.aw_size * 8 < 0x7FFFFFFFull && GL.ex_size * 8 < 0x7FFFFFFFull,
                  "oovqe_newton_direction: workspace per problem too large");
    const bool small_q = n - BW <= 64 * RQ_SMALL;
    const void* k1 = small_q ? (const void*)newton_band_kernel<RQ_SMALL> : (const void*)newton_band_kernel<RQ_LARGE>;
    OOVQE_CHECK_HIP(hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)newton_band_solve_kernel<false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    // workgroups per problem: all of them must be resident together (they wait for each other)
    const int lim1 = n2_resident_limit(k1, NT, lds);
    const int lim2 = n2_resident_limit((const void*)newton_band_solve_kernel<false>, NT2, lds2);
    const int cap1 = max_wg > 0 ? max_wg : 32, cap2 = max_wg > 0 ? max_wg : N2_MS_WGS;
    int W = 1;
    while (2 * W * batch <= lim1 && 2 * W <= cap1 && 2 * W <= GL.ntile) W *= 2;
    // exchange blocks and status words start as all ones: the pattern the hand-offs wait to see replaced
    OOVQE_CHECK_HIP(hipMemsetAsync(work + (size_t)batch * GL.aw_size, 0xFF,
                                   (size_t)batch * (GL.ex_size + 2) * sizeof(double), st),
                    "oovqe_newton_direction: memset");
    if (small_q)
        hipLaunchKernelGGL(newton_band_kernel<RQ_SMALL>, dim3(batch * W), dim3(NT), lds, st, hessian, gradient, n,
                           work, W, batch, (const double*)info, which);
    else
        hipLaunchKernelGGL(newton_band_kernel<RQ_LARGE>, dim3(batch * W), dim3(NT), lds, st, hessian, gradient, n,
                           work, W, batch, (const double*)info, which);
    OOVQE_CHECK_LAUNCH("oovqe_newton_direction/band");
    int W2 = 1;
    while (2 * W2 * batch <= lim2 && 2 * W2 <= cap2) W2 *= 2;
    hipLaunchKernelGGL(newton_band_solve_kernel<false>, dim3(batch * W2), dim3(NT2), lds2, st, n, lambda_min, mu,
                       rho, aug, work, dp, lowest_eigenvalue, shift, (double*)nullptr, W2, batch, info, which);
    OOVQE_CHECK_LAUNCH("oovqe_newton_direction/solve");
    return 0;
}

extern "C" int oovqe_newton_direction_pd(const double* hessian, const double* gradient, int n, int batch,
                                         double lambda_min, double* work, double* dp, double* shift, double* info,
                                         oovqe_stream_t stream)
{
    OOVQE_REQUIRE(hessian && gradient && work && dp && info, "oovqe_newton_direction_pd: null pointer");
    OOVQE_REQUIRE(batch >= 1 && batch <= 32767, "oovqe_newton_direction_pd: batch = %d", batch);
    return oovqe_newton_chol_launch(hessian, gradient, n, batch, lambda_min, work, dp, shift, info, (hipStream_t)stream);
}

extern "C" int oovqe_newton_direction_rest(const double* hessian, const double* gradient, int n, int batch,
                                           double lambda_min, double mu, double rho, int aug, double* info, int which,
                                           int max_wg, double* work, double* dp, double* lowest_eigenvalue,
                                           double* shift, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(hessian && gradient && work && dp && lowest_eigenvalue, "oovqe_newton_direction_rest: null pointer");
    OOVQE_REQUIRE(n >= 1 && n <= NEWTON3_NMAX, "oovqe_newton_direction_rest: n = %d outside 1..%d", n, NEWTON3_NMAX);
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "oovqe_newton_direction_rest: batch = %d", batch);
    OOVQE_REQUIRE(which >= 0 && which <= 2 && (which == 0 || info), "oovqe_newton_direction_rest: which = %d", which);
    hipStream_t st = (hipStream_t)stream;
    if (n > NEWTON2_NMAX) {
        OOVQE_REQUIRE(which == 0, "oovqe_newton_direction_rest: n = %d has no fast path to complement", n);
        return n3_direction(hessian, gradient, n, batch, lambda_min, mu, rho, aug, work, dp, lowest_eigenvalue,
                            shift, info, st);
    }
    if (n2_use_band(n, aug))
        return n2_band_route(hessian, gradient, n, batch, lambda_min, mu, rho, aug, work, dp, lowest_eigenvalue,
                             shift, info, which, max_wg, st);
    OOVQE_REQUIRE(which == 0, "oovqe_newton_direction_rest: the one-workgroup kernel serves every problem");
    OOVQE_REQUIRE(n <= NEWTON_NMAX, "oovqe_newton_direction: n = %d > %d needs the level shift (aug != 0)", n,
                  NEWTON_NMAX);
    const Layout L = make_layout(n);
    const size_t lds = (size_t)L.total * sizeof(double);
    OOVQE_REQUIRE(lds <= 160 * 1024, "oovqe_newton_direction: %zu bytes of LDS needed", lds);
    OOVQE_CHECK_HIP(hipFuncSetAttribute((const void*)newton_direction_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                    "oovqe_newton_direction: hipFuncSetAttribute");
    hipLaunchKernelGGL(newton_direction_kernel, dim3(batch), dim3(NT), lds, st, hessian,
                       gradient, n, lambda_min, mu, rho, aug, work, dp, lowest_eigenvalue, shift);
    OOVQE_CHECK_LAUNCH("oovqe_newton_direction");
    return 0;
}

extern "C" int oovqe_newton_direction(const double* hessian, const double* gradient, int n, int batch,
                                      double lambda_min, double mu, double rho, int aug, double* work,
                                      double* dp, double* lowest_eigenvalue, double* shift,
                                      oovqe_stream_t stream)
{
    OOVQE_REQUIRE(hessian && gradient && work && dp && lowest_eigenvalue, "oovqe_newton_direction: null pointer");
    OOVQE_REQUIRE(n >= 1 && n <= NEWTON3_NMAX, "oovqe_newton_direction: n = %d outside 1..%d", n, NEWTON3_NMAX);
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "oovqe_newton_direction: batch = %d", batch);
    if (n2_use_chol(n, aug) && batch <= 32767) {
        // positive-definite Hessians: direction from the Cholesky factorisation, the band route only adds the
        // lowest eigenvalue; the others go through the band route entirely
        double* cw = work + n2_route_work(n, batch);
        double* info = cw + oovqe_newton_chol_work(n, batch);
        int rc = oovqe_newton_direction_pd(hessian, gradient, n, batch, lambda_min, cw, dp, shift, info, stream);
        if (rc) return rc;
        return oovqe_newton_direction_rest(hessian, gradient, n, batch, lambda_min, mu, rho, aug, info, 0, 0, work, dp,
                                           lowest_eigenvalue, shift, stream);
    }
    return oovqe_newton_direction_rest(hessian, gradient, n, batch, lambda_min, mu, rho, aug, nullptr, 0, 0, work, dp,
                                       lowest_eigenvalue, shift, stream);
}
