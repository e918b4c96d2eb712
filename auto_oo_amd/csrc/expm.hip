// K2: matrix exponential U = expm(sign * X) by scaling-and-squaring with a Taylor polynomial of
// degree 8, 12 or 16 (by the scaled norm) evaluated in Paterson-Stockmeyer form (4, 5 or 6 matrix
// products + s squarings, all on the fp64 MFMA).  Replaces math.expm(-K) = torch.linalg.matrix_exp (reference
// src/auto_oo/oo_energy.py:226-230) together with the kappa -> skew-matrix scatter
// (oo_energy.py:63-87,213-219).
//
// With ||A||_1 <= 1/2 after scaling the truncation error of degree 16 is 0.5^17/17! = 2e-20 (degree
// 12 up to 0.3: 3e-17, degree 8 up to 0.06: 3e-17), i.e. the result is accurate to fp64 rounding (times the 2^s growth of the squarings), independent of the
// reference's own (Pade / Taylor) degree choice.
//
//   N <= 48 : ONE workgroup, all six N x N matrices resident in LDS (<= 115 KiB), the number
//             of squarings decided on the device - a single launch, no host round trip.
//   N  > 48 : matrices in HBM/L2, products through the K1 contraction kernel; the 1-norm is read
//             back once (8 bytes) to choose s on the host.
#include "common.h"

int oovqe_mode_contract_impl(const double* T, const double* Cm, double* out, long A, int K, int J,
                             long B, int ldc, int last, hipStream_t st);

namespace {

constexpr double THETA = 0.5;
// Taylor degree by the scaled 1-norm t: the remainder t^(m+1)/(m+1)! stays below 3e-17 for degree 8
// up to 0.06, degree 12 up to 0.3, degree 16 up to 0.5 -- one / two / three Horner products with A^4
// after the three products that form A^2, A^3, A^4 (orbital-rotation steps are usually small)
constexpr double THETA8 = 0.06, THETA12 = 0.3;
constexpr int SMALL_MAX = 48;
constexpr int EX_THREADS = 512;

// 1/k!
__constant__ double INV_FACT[17] = {
    1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
    1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
    1.0 / 1307674368000.0, 1.0 / 20922789888000.0};
static const double H_INV_FACT[17] = {
    1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
    1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
    1.0 / 1307674368000.0, 1.0 / 20922789888000.0};

// C = A * B for zero-padded [NP][LD] LDS matrices (NP = 16 * nt).  8 waves share the nt^2 tiles.
__device__ void lds_matmul(const double* A, const double* B, double* C, int nt, int ksteps, int LD)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    for (int tile = wave; tile < nt * nt; tile += EX_THREADS / 64) {
        const int m0 = (tile / nt) * 16, n0 = (tile % nt) * 16;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* ap = A + (m0 + lr) * LD + lq;
        const double* bp = B + lq * LD + n0 + lr;
        for (int ks = 0; ks < ksteps; ++ks) acc = mfma_f64(ap[ks * 4], bp[ks * 4 * LD], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) C[(m0 + lq + 4 * i) * LD + n0 + lr] = acc[i];
    }
    __syncthreads();
}

// P = alpha*T + c0 I + c1 A + c2 A2 + c3 A3  on the N x N block
__device__ void lds_combine(double* P, const double* T, double alpha, const double* A,
                            const double* A2, const double* A3, double c0, double c1, double c2,
                            double c3, int N, int LD)
{
    for (int idx = threadIdx.x; idx < N * N; idx += EX_THREADS) {
        const int r = idx / N, c = idx - r * N;
        const int o = r * LD + c;
        double v = c1 * A[o] + c2 * A2[o] + c3 * A3[o] + (r == c ? c0 : 0.0);
        if (T) v += alpha * T[o];
        P[o] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(EX_THREADS)
void expm_small_kernel(const double* __restrict__ X, const double* __restrict__ kappa,
                       const int32_t* __restrict__ kap_row, const int32_t* __restrict__ kap_col,
                       int n_kappa, double sign, int N, double* __restrict__ Kout,
                       double* __restrict__ U, const double* __restrict__ Cin, double* __restrict__ Cout)
{
    // blockIdx.x = element of a batch: every array advances by its own size
    {
        const size_t b = blockIdx.x, n2 = (size_t)N * N;
        if (X) X += b * n2;
        if (kappa) kappa += b * n_kappa;
        if (Kout) Kout += b * n2;
        if (U) U += b * n2;
        if (Cin) Cin += b * n2;
        if (Cout) Cout += b * n2;
    }
    extern __shared__ double lds[];
    __shared__ double colsum[SMALL_MAX];
    __shared__ int s_shared, m_shared;
    const int nt = (N + 15) / 16, NP = nt * 16;
    const int LD = NP + 2;
    const int msz = NP * LD;
    double* A = lds;
    double* A2 = A + msz;
    double* A3 = A2 + msz;
    double* A4 = A3 + msz;
    double* P = A4 + msz;
    double* T = P + msz;
    const int tid = threadIdx.x;
    const int ksteps = (N + 3) / 4;

    for (int idx = tid; idx < 6 * msz; idx += EX_THREADS) lds[idx] = 0.0;
    __syncthreads();
    if (X) {
        for (int idx = tid; idx < N * N; idx += EX_THREADS) {
            const int r = idx / N, c = idx - r * N;
            A[r * LD + c] = sign * X[idx];
        }
    } else {
        // kappa vector -> skew matrix K (lower = +kappa, upper = -kappa); A = sign * K
        for (int t = tid; t < n_kappa; t += EX_THREADS) {
            const int r = kap_row[t], c = kap_col[t];
            const double v = kappa[t];
            A[r * LD + c] = sign * v;
            A[c * LD + r] = -sign * v;
        }
    }
    __syncthreads();
    if (Kout) {
        for (int idx = tid; idx < N * N; idx += EX_THREADS) {
            const int r = idx / N, c = idx - r * N;
            Kout[idx] = sign * A[r * LD + c];   // sign = +-1: undo
        }
    }
    // 1-norm and scaling
    if (tid < N) {
        double cs = 0.0;
        for (int r = 0; r < N; ++r) cs += fabs(A[r * LD + tid]);
        colsum[tid] = cs;
    }
    __syncthreads();
    if (tid == 0) {
        double nrm = 0.0;
        for (int c = 0; c < N; ++c) nrm = fmax(nrm, colsum[c]);
        int s = 0;
        while (nrm > THETA && s < 60) { nrm *= 0.5; ++s; }
        s_shared = s;
        m_shared = nrm <= THETA8 ? 8 : nrm <= THETA12 ? 12 : 16;
    }
    __syncthreads();
    const int s = s_shared, mdeg = m_shared;
    const double scale = ldexp(1.0, -s);
    for (int idx = tid; idx < N * N; idx += EX_THREADS) {
        const int r = idx / N, c = idx - r * N;
        A[r * LD + c] *= scale;
    }
    __syncthreads();
    lds_matmul(A, A, A2, nt, ksteps, LD);
    lds_matmul(A2, A, A3, nt, ksteps, LD);
    lds_matmul(A2, A2, A4, nt, ksteps, LD);
    const double* f = INV_FACT;
    // top block: P = c_m A4 + (c_{m-4} I + c_{m-3} A + c_{m-2} A2 + c_{m-1} A3), then Horner in A4
    lds_combine(P, A4, f[mdeg], A, A2, A3, f[mdeg - 4], f[mdeg - 3], f[mdeg - 2], f[mdeg - 1], N, LD);
    for (int b0 = mdeg - 8; b0 >= 0; b0 -= 4) {
        lds_matmul(P, A4, T, nt, ksteps, LD);
        lds_combine(P, T, 1.0, A, A2, A3, f[b0], f[b0 + 1], f[b0 + 2], f[b0 + 3], N, LD);
    }
    double* cur = P;
    double* oth = T;
    for (int i = 0; i < s; ++i) {
        lds_matmul(cur, cur, oth, nt, ksteps, LD);
        double* tmp = cur; cur = oth; oth = tmp;
    }
    if (U)
        for (int idx = tid; idx < N * N; idx += EX_THREADS) {
            const int r = idx / N, c = idx - r * N;
            U[idx] = cur[r * LD + c];
        }
    if (Cout) {
        // rotated orbitals Cout = Cin expm(sign X) (oo_energy.py:232-236) without leaving LDS
        for (int idx = tid; idx < N * N; idx += EX_THREADS) {
            const int r = idx / N, c = idx - r * N;
            A[r * LD + c] = Cin[idx];
        }
        __syncthreads();
        lds_matmul(A, cur, oth, nt, ksteps, LD);
        for (int idx = tid; idx < N * N; idx += EX_THREADS) {
            const int r = idx / N, c = idx - r * N;
            Cout[idx] = oth[r * LD + c];
        }
    }
}

// ---- large-N helpers (matrices in global memory, [N][N] dense) ------------------------------
__global__ void scatter_skew_kernel(const double* kappa, const int32_t* kap_row,
                                    const int32_t* kap_col, int n_kappa, int N, double* K)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_kappa) {
        const int r = kap_row[t], c = kap_col[t];
        K[(size_t)r * N + c] = kappa[t];
        K[(size_t)c * N + r] = -kappa[t];
    }
}

// 1-norm (largest absolute column sum) in two small launches: 16 row slices per 64 columns, then one
// workgroup folds the slices and takes the maximum (one workgroup walking all N rows of its columns
// took 35 us at N = 200, 140 us at N = 300: a chain of N dependent L2 round trips).
constexpr int NORM_SLICES = 16;

__global__ __launch_bounds__(64)
void norm1_partial_kernel(const double* __restrict__ X, int N, double* __restrict__ part)
{
    const int c = blockIdx.x * 64 + threadIdx.x, slice = blockIdx.y;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = slice;
    for (; r + 3 * NORM_SLICES < N; r += 4 * NORM_SLICES) {
        s0 += fabs(X[(size_t)r * N + c]);
        s1 += fabs(X[(size_t)(r + NORM_SLICES) * N + c]);
        s2 += fabs(X[(size_t)(r + 2 * NORM_SLICES) * N + c]);
        s3 += fabs(X[(size_t)(r + 3 * NORM_SLICES) * N + c]);
    }
    for (; r < N; r += NORM_SLICES) s0 += fabs(X[(size_t)r * N + c]);
    part[(size_t)slice * N + c] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256)
void norm1_kernel(const double* __restrict__ part, int N, double* out)
{
    __shared__ double red[256];
    double best = 0.0;
    for (int c = threadIdx.x; c < N; c += 256) {
        double cs = 0.0;
#pragma unroll
        for (int sl = 0; sl < NORM_SLICES; ++sl) cs += part[(size_t)sl * N + c];
        best = fmax(best, cs);
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void scale_kernel(const double* X, double f, long n, double* A)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i] = f * X[i];
}

__global__ void combine_kernel(double* P, const double* T, double alpha, const double* A,
                               const double* A2, const double* A3, double c0, double c1, double c2,
                               double c3, int N)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * N) return;
    const int r = (int)(i / N), c = (int)(i - (long)r * N);
    double v = c1 * A[i] + c2 * A2[i] + c3 * A3[i] + (r == c ? c0 : 0.0);
    if (T) v += alpha * T[i];
    P[i] = v;
}

int expm_large(const double* X, double sign, int N, double* U, double* work, hipStream_t st)
{
    const long n2 = (long)N * N;
    double* A = work;
    double* A2 = A + n2;
    double* A3 = A2 + n2;
    double* A4 = A3 + n2;
    double* P = A4 + n2;
    double* T = P + n2;
    // 1-norm -> host (8 bytes) -> number of squarings
    // (partial sums in P, [16][N] <= N^2 for N >= 16; smaller matrices take the one-workgroup kernel)
    norm1_partial_kernel<<<dim3((N + 63) / 64, NORM_SLICES), 64, 0, st>>>(X, N, P);
    norm1_kernel<<<1, 256, 0, st>>>(P, N, T);
    double nrm = 0.0;
    hipError_t e = hipMemcpyAsync(&nrm, T, sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        oovqe_set_error("expm: norm readback: %s", hipGetErrorString(e));
        return OOVQE_ERR_HIP;
    }
    if (!(nrm == nrm) || nrm > 1e300) {
        oovqe_set_error("expm: non-finite input");
        return OOVQE_ERR_ARG;
    }
    int s = 0;
    while (nrm > THETA && s < 60) { nrm *= 0.5; ++s; }
    const int mdeg = nrm <= THETA8 ? 8 : nrm <= THETA12 ? 12 : 16;
    const unsigned nb = (unsigned)((n2 + 255) / 256);
    scale_kernel<<<nb, 256, 0, st>>>(X, sign * ldexp(1.0, -s), n2, A);
    int rc;
#define MM(a, b, c) if ((rc = oovqe_mode_contract_impl(a, b, c, N, N, N, 1, N, 1, st))) return rc
    MM(A, A, A2);
    MM(A2, A, A3);
    MM(A2, A2, A4);
    const double* f = H_INV_FACT;
    combine_kernel<<<nb, 256, 0, st>>>(P, A4, f[mdeg], A, A2, A3, f[mdeg - 4], f[mdeg - 3], f[mdeg - 2],
                                       f[mdeg - 1], N);
    for (int b0 = mdeg - 8; b0 >= 0; b0 -= 4) {
        MM(P, A4, T);
        combine_kernel<<<nb, 256, 0, st>>>(P, T, 1.0, A, A2, A3, f[b0], f[b0 + 1], f[b0 + 2], f[b0 + 3], N);
    }
    double* cur = P;
    double* oth = T;
    for (int i = 0; i < s; ++i) {
        MM(cur, cur, oth);
        double* tmp = cur; cur = oth; oth = tmp;
    }
#undef MM
    e = hipMemcpyAsync(U, cur, n2 * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        oovqe_set_error("expm: copy: %s", hipGetErrorString(e));
        return OOVQE_ERR_HIP;
    }
    OOVQE_CHECK_LAUNCH("expm");
    return 0;
}

int expm_small(const double* X, const double* kappa, const int32_t* kap_row, const int32_t* kap_col,
               int n_kappa, double sign, int N, double* Kout, double* U, hipStream_t st, int batch = 1,
               const double* Cin = nullptr, double* Cout = nullptr)
{
    const int nt = (N + 15) / 16, NP = nt * 16, LD = NP + 2;
    const size_t lds_bytes = (size_t)6 * NP * LD * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)expm_small_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) {
            oovqe_set_error("expm: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(expm_small_kernel, dim3(batch), dim3(EX_THREADS), lds_bytes, st, X, kappa,
                       kap_row, kap_col, n_kappa, sign, N, Kout, U, Cin, Cout);
    OOVQE_CHECK_LAUNCH("expm_small");
    return 0;
}

}  // namespace

extern "C" int oovqe_expm(const double* X, double sign, int N, double* U, double* work,
                          oovqe_stream_t stream)
{
    OOVQE_REQUIRE(X && U, "expm: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 4096, "expm: N=%d", N);
    OOVQE_REQUIRE(sign == 1.0 || sign == -1.0, "expm: sign must be +-1");
    hipStream_t st = (hipStream_t)stream;
    if (N <= SMALL_MAX) return expm_small(X, nullptr, nullptr, nullptr, 0, sign, N, nullptr, U, st);
    OOVQE_REQUIRE(work, "expm: work buffer required for N > %d", SMALL_MAX);
    return expm_large(X, sign, N, U, work, st);
}

extern "C" int oovqe_expm_skew(const double* kappa, const int32_t* kap_row, const int32_t* kap_col,
                               int n_kappa, int N, double* K, double* U, double* work,
                               oovqe_stream_t stream)
{
    OOVQE_REQUIRE(kappa && kap_row && kap_col && U, "expm_skew: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 4096 && n_kappa >= 0, "expm_skew: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (N <= SMALL_MAX)
        return expm_small(nullptr, kappa, kap_row, kap_col, n_kappa, -1.0, N, K, U, st);
    OOVQE_REQUIRE(work, "expm_skew: work buffer (7*N*N doubles) required for N > %d", SMALL_MAX);
    // K lives in the 7th N*N block of work unless the caller wants it back
    double* Kbuf = K ? K : work + (size_t)6 * N * N;
    hipError_t e = hipMemsetAsync(Kbuf, 0, (size_t)N * N * sizeof(double), st);
    if (e != hipSuccess) {
        oovqe_set_error("expm_skew: memset: %s", hipGetErrorString(e));
        return OOVQE_ERR_HIP;
    }
    if (n_kappa > 0)
        scatter_skew_kernel<<<(n_kappa + 255) / 256, 256, 0, st>>>(kappa, kap_row, kap_col, n_kappa,
                                                                  N, Kbuf);
    return expm_large(Kbuf, -1.0, N, U, work, st);
}

// Orbital rotation of a stack of geometries in ONE launch (OO_energy.get_transformed_mo,
// src/auto_oo/oo_energy.py:213-236, per geometry): C_out[b] = C[b] expm(-K(kappa[b])).
// kappa [batch][n_kappa], C / C_out [batch][N][N] (C_out may alias C), U [batch][N][N] or NULL.
// N <= 48: one workgroup per geometry, everything in LDS.
int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);

extern "C" int oovqe_rotate_orbitals_batch(const double* kappa, const int32_t* kap_row,
                                           const int32_t* kap_col, int n_kappa, int N, int batch,
                                           const double* C, double* C_out, double* U, double* work,
                                           oovqe_stream_t stream)
{
    OOVQE_REQUIRE(kappa && kap_row && kap_col && C && C_out, "rotate_orbitals_batch: null pointer");
    OOVQE_REQUIRE(N >= 1 && N <= 4096 && n_kappa >= 0 && batch >= 1 && batch <= 65535,
                  "rotate_orbitals_batch: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (N <= SMALL_MAX)
        return expm_small(nullptr, kappa, kap_row, kap_col, n_kappa, -1.0, N, nullptr, U, st, batch, C, C_out);
    // larger orbital spaces: one expm per geometry through the multi-launch path, then one batched product
    OOVQE_REQUIRE(work, "rotate_orbitals_batch: work (batch*N*N + 7*N*N doubles) required for N > %d", SMALL_MAX);
    OOVQE_REQUIRE(C_out != C, "rotate_orbitals_batch: C_out must not alias C for N > %d", SMALL_MAX);
    const size_t n2 = (size_t)N * N;
    double* Ub = U ? U : work;                 // [batch][N][N]
    double* ew = work + (size_t)batch * n2;    // 7 N^2 scratch of oovqe_expm_skew
    int rc;
    for (int b = 0; b < batch; ++b)
        if ((rc = oovqe_expm_skew(kappa + (size_t)b * n_kappa, kap_row, kap_col, n_kappa, N, nullptr,
                                  Ub + (size_t)b * n2, ew, stream)))
            return rc;
    // C_out[b] = C[b] U[b]  (LAST-mode contraction: T = C [N,N], Cm = U)
    return oovqe_mode_contract_batched(C, Ub, C_out, N, N, N, 1, N, 1, batch, (long)n2, (long)n2, (long)n2, st);
}
