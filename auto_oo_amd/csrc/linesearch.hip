// Book-keeping of the batched backtracking line search (newton_raphson.py:131-192 of the reference, for G
// problems in lockstep) as two small kernels, so that a trial costs the host two launches and one 32-byte
// readback instead of ~20 elementwise torch launches:
//   oovqe_linesearch_points   points = flat + t dp (split into the theta and the kappa part), and on the first
//                             call the Armijo slope alpha <grad, dp> of every problem;
//   oovqe_linesearch_update   the reference's acceptance rule per problem (newton_raphson.py:146-177: a trial
//                             passes when E_trial <= E + t slope; a NaN never passes), the next step length of
//                             the problems still searching, and the numbers the host decides on.
#include "common.h"

namespace {

__device__ __forceinline__ double ls_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one workgroup of 256 threads per problem
__global__ __launch_bounds__(256)
void linesearch_points_kernel(const double* __restrict__ flat, const double* __restrict__ dp,
                              const double* __restrict__ t, const double* __restrict__ grad, double alpha, int n,
                              int n_a, double* __restrict__ pa, double* __restrict__ pb,
                              double* __restrict__ slope)
{
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double tb = t[b];
    const double* f = flat + (size_t)b * n;
    const double* d = dp + (size_t)b * n;
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double di = d[i];
        const double v = fma(tb, di, f[i]);
        if (i < n_a) pa[(size_t)b * n_a + i] = v;
        else pb[(size_t)b * (n - n_a) + i - n_a] = v;
        if (slope) acc = fma(grad[(size_t)b * n + i], di, acc);
    }
    if (slope) {
        acc = ls_wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) slope[b] = alpha * (red[0] + red[1] + red[2] + red[3]);
    }
}

// one workgroup for all problems: flags = [any problem still searching, min info, any NaN slope,
// any searching problem whose slope is not negative]
__global__ __launch_bounds__(256)
void linesearch_update_kernel(const double* __restrict__ trial, long trial_stride, const double* __restrict__ energy,
                              const double* __restrict__ slope, const double* __restrict__ info, double beta,
                              int first, int give_up, int batch, double* __restrict__ t, double* __restrict__ active,
                              double* __restrict__ best, double* __restrict__ flags)
{
    __shared__ double s_any, s_min, s_nan, s_up;
    const int tid = threadIdx.x;
    if (tid == 0) { s_any = 0.0; s_min = 1.0e300; s_nan = 0.0; s_up = 0.0; }
    __syncthreads();
    double l_any = 0.0, l_min = 1.0e300, l_nan = 0.0, l_up = 0.0;
    for (int b = tid; b < batch; b += 256) {
        const double e0 = energy[b], sl = slope[b];
        double tb = t[b];
        const bool was = first ? true : active[b] != 0.0;
        bool act = false;
        if (was) {
            if (give_up) {
                // newton_raphson.py:177-183: after lmax + 1 reductions the search gives up and keeps the old
                // parameters (whatever the last trial said)
                tb = 0.0;
                best[b] = e0;
            } else {
                const double tr = trial[(size_t)b * trial_stride];
                act = !(tr <= e0 + tb * sl);                 // (a NaN trial never passes)
                if (!act) best[b] = tr;
                else tb *= beta;                             // the next trial of this problem
            }
            t[b] = tb;
        }
        active[b] = act ? 1.0 : 0.0;
        if (act) { l_any = 1.0; if (!(sl < 0.0)) l_up = 1.0; }
        if (sl != sl) l_nan = 1.0;
        if (info) l_min = fmin(l_min, info[b]);
    }
    // (tiny reduction: a few atomics in LDS would do; keep it simple and deterministic)
    for (int o = 32; o > 0; o >>= 1) {
        l_any = fmax(l_any, __shfl_xor(l_any, o, 64));
        l_nan = fmax(l_nan, __shfl_xor(l_nan, o, 64));
        l_up = fmax(l_up, __shfl_xor(l_up, o, 64));
        l_min = fmin(l_min, __shfl_xor(l_min, o, 64));
    }
    __shared__ double r_any[4], r_min[4], r_nan[4], r_up[4];
    if ((tid & 63) == 0) { r_any[tid >> 6] = l_any; r_min[tid >> 6] = l_min; r_nan[tid >> 6] = l_nan; r_up[tid >> 6] = l_up; }
    __syncthreads();
    if (tid == 0) {
        flags[0] = fmax(fmax(r_any[0], r_any[1]), fmax(r_any[2], r_any[3]));
        flags[1] = info ? fmin(fmin(r_min[0], r_min[1]), fmin(r_min[2], r_min[3])) : 0.0;
        flags[2] = fmax(fmax(r_nan[0], r_nan[1]), fmax(r_nan[2], r_nan[3]));
        flags[3] = fmax(fmax(r_up[0], r_up[1]), fmax(r_up[2], r_up[3]));
    }
}

}  // namespace

extern "C" int oovqe_linesearch_points(const double* flat, const double* dp, const double* t, const double* grad,
                                       double alpha, int n, int n_a, int batch, double* points_a, double* points_b,
                                       double* slope, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(flat && dp && t && points_a, "linesearch_points: null pointer");
    OOVQE_REQUIRE(n >= 1 && n_a >= 0 && n_a <= n && (n_a == n || points_b) && batch >= 1 && batch <= 65535,
                  "linesearch_points: bad sizes");
    OOVQE_REQUIRE(!slope || grad, "linesearch_points: the slope needs the gradient");
    if (n_a == 0) { n_a = n; }      // no split: everything into points_a
    hipLaunchKernelGGL(linesearch_points_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, flat, dp, t, grad,
                       alpha, n, n_a, points_a, points_b, slope);
    OOVQE_CHECK_LAUNCH("oovqe_linesearch_points");
    return 0;
}

extern "C" int oovqe_linesearch_update(const double* trial, int64_t trial_stride, const double* energy,
                                       const double* slope, const double* info, double beta, int first, int give_up,
                                       int batch, double* t, double* active, double* best, double* flags,
                                       oovqe_stream_t stream)
{
    OOVQE_REQUIRE(energy && slope && t && active && best && flags && (trial || give_up), "linesearch_update: null pointer");
    OOVQE_REQUIRE(batch >= 1, "linesearch_update: batch = %d", batch);
    hipLaunchKernelGGL(linesearch_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, trial, (long)trial_stride,
                       energy, slope, info, beta, first, give_up, batch, t, active, best, flags);
    OOVQE_CHECK_LAUNCH("oovqe_linesearch_update");
    return 0;
}
