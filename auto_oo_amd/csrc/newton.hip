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

namespace {

constexpr int NT = 1024;        // threads per workgroup (thread i <-> row i of the problem)
constexpr int NW = NT / 64;     // waves
constexpr int NB = 16;          // panel width
constexpr int NEWTON_NMAX = 480;

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

struct Layout {     // dynamic LDS, in doubles
    int npv;        // pitch of Vc / Wc rows, >= n + 16, multiple of 16
    int npb;        // length of the v vector, >= n + 128
    int Vc, Wc, vb, pb, x1, x2, Tl, red, dd, ee, bb, aux, total;
};

__host__ __device__ inline Layout make_layout(int n)
{
    Layout L;
    L.npv = ((n + 16 + 15) / 16) * 16;
    L.npb = ((n + 128 + 15) / 16) * 16;
    int o = 0;
    L.Vc = o; o += NB * L.npv;
    L.Wc = o; o += NB * L.npv;
    L.vb = o; o += L.npb;
    L.pb = o; o += L.npv;
    L.x1 = o; o += NB;
    L.x2 = o; o += NB;
    L.Tl = o; o += NB * NB;
    L.red = o; o += 2 * 2 * NW;
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

    // working copy of H, rows padded to an even pitch (pad column = 0)
    for (int idx = tid; idx < n * lda; idx += NT) {
        const int r = idx / lda, c = idx - r * lda;
        Aw[idx] = c < n ? Hb[(size_t)r * n + c] : 0.0;
    }
    for (int idx = tid; idx < L.npb; idx += NT) vb[idx] = 0.0;
    __syncthreads();

    const int i = tid;          // this thread's row
    for (int pi = 0; pi < npan; ++pi) {
        const int k0 = pi * NB;
        const int jb = (n - 1 - k0) < NB ? (n - 1 - k0) : NB;
        for (int idx = tid; idx < NB * npv; idx += NT) { Vc[idx] = 0.0; Wc[idx] = 0.0; }
        if (tid < NB * NB) Tl[tid] = 0.0;
        // a short last panel: its unused reflector rows must read as zero in the Q application
        for (int idx = jb * npv + tid; idx < NB * npv; idx += NT) Vst[(size_t)pi * NB * npv + idx] = 0.0;
        __syncthreads();
        for (int j = 0; j < jb; ++j) {
            const int c = k0 + j;
            // ---- column c of the matrix with the panel's pending rank-2 updates applied
            double colv = 0.0;
            if (i >= c && i < n) {
                colv = Aw[(size_t)c * lda + i];        // row c == column c (full storage is kept symmetric)
                for (int l = 0; l < j; ++l)
                    colv -= Vc[l * npv + i] * Wc[l * npv + c] + Wc[l * npv + i] * Vc[l * npv + c];
            }
            if (i == c) my_d = colv;
            double s2 = (i > c + 1 && i < n) ? colv * colv : 0.0;
            double al = (i == c + 1) ? colv : 0.0;
            block_sum2(s2, al, red, parity, lane, wave);
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
            __syncthreads();
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
                if (lane == 0) { x1[wave] = a1; x2[wave] = a2; }
            }
            // ---- p_raw = A22 v: the trailing matrix as of the panel's start, streamed from L2;
            // a wave takes four rows at a time, its lanes run along the columns with 16-byte loads
            {
                const int cs = (c + 1) & ~1;
                const int m = n - (c + 1);
                const int nch = (n - cs + 127) >> 7;          // 128-column chunks, <= 4
                for (int g0 = wave; 4 * g0 < m; g0 += NW) {
                    const int rbase = c + 1 + 4 * g0;
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
                    const double* rp[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = rbase + rr < n ? rbase + rr : n - 1;
                        rp[rr] = Aw + (size_t)r * lda;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t < nch) {
                            const int col = cs + 2 * lane + 128 * t;
                            if (col < n) {
                                const d2 vv = *reinterpret_cast<const d2*>(vb + col);
#pragma unroll
                                for (int rr = 0; rr < 4; ++rr) {
                                    const d2 a = *reinterpret_cast<const d2*>(rp[rr] + col);
                                    acc[rr] += a.x * vv.x + a.y * vv.y;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const double s = wave_sum(acc[rr]);
                        if (lane == 0 && rbase + rr < n) pb[rbase + rr] = s;
                    }
                }
            }
            __syncthreads();
            // ---- p = tau (A22 v - V (W^T v) - W (V^T v)),  w = p - (tau/2)(p^T v) v
            double pi_ = 0.0;
            if (i > c && i < n) {
                double s = pb[i];
                for (int l = 0; l < j; ++l) s -= Vc[l * npv + i] * x1[l] + Wc[l * npv + i] * x2[l];
                pi_ = tau * s;
            }
            double pv = pi_ * vi, zero = 0.0;
            block_sum2(pv, zero, red, parity, lane, wave);
            const double wi = pi_ - 0.5 * tau * pv * vi;
            if (i < npv) Wc[j * npv + i] = wi;
            // compact WY factor of the panel: Q_panel = I - V T V^T, T upper triangular
            if (tid < j) {
                double s = 0.0;
                for (int mm = 0; mm < j; ++mm) s += Tl[tid * NB + mm] * x2[mm];
                Tl[tid * NB + j] = -tau * s;
            } else if (tid == j) {
                Tl[j * NB + j] = tau;
            }
            __syncthreads();
        }
        // ---- trailing matrix -= V W^T + W V^T (rows and columns from r0 on), 16x16 tiles on the
        // fp64 matrix cores: lane supplies A[m = lane&15][k = lane>>4], B[k][n = lane&15]
        {
            const int r0 = k0 + jb;
            const int mt = (n - r0 + 15) / 16;
            const int lq = lane >> 4, lr = lane & 15;
            for (int tile = wave; tile < mt * mt; tile += NW) {
                const int ti = tile / mt, tj = tile - ti * mt;
                const int rowb = r0 + 16 * ti, colb = r0 + 16 * tj;
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = lq + 4 * s;
                    acc = mfma_f64(Vc[k * npv + rowb + lr], Wc[k * npv + colb + lr], acc);
                    acc = mfma_f64(Wc[k * npv + rowb + lr], Vc[k * npv + colb + lr], acc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = rowb + lq + 4 * e, col = colb + lr;
                    if (row < n && col < n) Aw[(size_t)row * lda + col] -= acc[e];
                }
            }
            if (tid < NB * NB) Tst[(size_t)pi * NB * NB + tid] = Tl[tid];
        }
        __syncthreads();
    }
    if (n >= 1 && i == n - 1) my_d = Aw[(size_t)(n - 1) * lda + (n - 1)];

    // ---------------- tridiagonal phase ----------------
    double* dd = sm + L.dd;
    double* ee = sm + L.ee;
    double* bb = sm + L.bb;
    double* aux = sm + L.aux;
    if (i < n) { dd[i] = my_d; ee[i] = (i < n - 1) ? my_e : 0.0; bb[i] = -gb[i]; }
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
        // Sturm count: number of eigenvalues of T below x (only "is there one" is needed)
        bool below = false;
        double q = dd[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        below = q < 0.0;
        for (int k = 1; k < n && !below; ++k) {
            const double e = ee[k - 1];
            q = dd[k] - x - e * e / q;
            if (fabs(q) < pivmin) q = -pivmin;
            below = q < 0.0;
        }
        double first = below ? (double)tid : (double)NT, dum = 0.0;
        block_min2(first, dum, red, parity, lane, wave);
        const int t0 = (int)first;            // first shift with an eigenvalue below it (NT: none, then hi stays)
        const double nlo = t0 > 0 ? lo + width * ((double)t0 / (double)(NT + 1)) : lo;
        const double nhi = t0 < NT ? lo + width * ((double)(t0 + 1) / (double)(NT + 1)) : hi;
        lo = nlo; hi = nhi;
    }
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
    // ---- (T + nu I) y = b: Gaussian elimination with partial pivoting on the tridiagonal matrix
    if (tid == 0) {
        double* dl = aux;            // sub-diagonal, then second super-diagonal
        double* du = aux + n;        // super-diagonal
        for (int k = 0; k < n; ++k) { dd[k] += nu; dl[k] = ee[k]; du[k] = ee[k]; }
        for (int k = 0; k + 1 < n; ++k) {
            const bool lastk = k + 2 >= n;
            if (fabs(dd[k]) >= fabs(dl[k])) {
                const double fact = dl[k] / dd[k];
                dd[k + 1] -= fact * du[k];
                bb[k + 1] -= fact * bb[k];
                dl[k] = 0.0;
            } else {
                const double fact = dd[k] / dl[k];
                dd[k] = dl[k];
                double temp = dd[k + 1];
                dd[k + 1] = du[k] - fact * temp;
                if (!lastk) { dl[k] = du[k + 1]; du[k + 1] = -fact * dl[k]; } else dl[k] = 0.0;
                du[k] = temp;
                temp = bb[k];
                bb[k] = bb[k + 1];
                bb[k + 1] = temp - fact * bb[k + 1];
            }
        }
        bb[n - 1] = bb[n - 1] / dd[n - 1];
        if (n > 1) bb[n - 2] = (bb[n - 2] - du[n - 2] * bb[n - 1]) / dd[n - 2];
        for (int k = n - 3; k >= 0; --k) bb[k] = (bb[k] - du[k] * bb[k + 1] - dl[k] * bb[k + 2]) / dd[k];
    }
    __syncthreads();
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
