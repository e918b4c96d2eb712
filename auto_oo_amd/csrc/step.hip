// One damped Newton step of a STACK of geometries in lockstep, enqueued by one call
// (the body of OO_pqc.full_optimization / of the Berry-phase loop, src/auto_oo/oo_pqc.py:172-196 and
// examples/Tutorial_Berry_phase.ipynb raw 408-441, per geometry; utils/newton_raphson.py:78-211).
//
// Why a call of its own: at the 8 geometries a rank of an 8-GPU job holds, the step is a chain of ~45
// latency-bound launches, and driving that chain from the host language costs as much as a fifth of it
// (the gradient / energy / parameter rows were torch copies and fills, the direction's entry points and
// the line search's first trial separate library calls with host work between them).  Here the whole chain
// up to the FIRST verdict of the line search is put on the stream back to back:
//
//   oovqe_oo_hessian_batch          E, full gradient, full Hessian of every geometry
//   step_prepare_kernel             gradient rows / energies made contiguous, flat = [theta | 0], t = 1
//   oovqe_newton_direction_pd       Cholesky fast path (info = 1 where it served)
//   oovqe_newton_direction_rest     the other problems: on the calling stream -- or, `speculate`, on the side
//                                   stream with the eigenvalues (the caller then checks flags[1])
//   side stream                     lowest eigenvalues of the positive definite Hessians (a reported number
//                                   no step reads, newton_raphson.py:105-128)
//   oovqe_linesearch_points         points = flat + dp, Armijo slopes
//   oovqe_rotate_orbitals_batch     trial orbitals C_oao expm(-K) ...
//   oovqe_matmul_nn_batch           ... and S^-1/2 (C_oao U)            (oo_pqc.py:191, oo_energy.py:173-176)
//   oovqe_oo_eval_batch             energies at the trial points
//   oovqe_linesearch_update         the acceptance rule per problem (newton_raphson.py:146-177), flags [4]
//
// The host reads the 32 bytes of `flags` and -- in the common case, every problem accepted at t = 1 -- is done:
// the trial's orbitals are the new orbitals.  Further trials (rare) go through the single entry points.
// Every launch is the one the separate calls make: the numbers are bit for bit those of the host-driven step.
#include "common.h"

namespace {

// one workgroup per geometry: grad[b,:] = out[b][2 : 2 + n], energy[b] = out[b][1], flat[b,:] = [theta[b,:] | 0],
// t[b] = 1
__global__ __launch_bounds__(256)
void step_prepare_kernel(const double* __restrict__ out, long out_stride, const double* __restrict__ theta,
                         int n_theta, int n, double* __restrict__ grad, double* __restrict__ energy,
                         double* __restrict__ flat, double* __restrict__ t)
{
    const int b = blockIdx.x;
    const double* o = out + (size_t)b * out_stride;
    for (int i = threadIdx.x; i < n; i += 256) {
        grad[(size_t)b * n + i] = o[2 + i];
        flat[(size_t)b * n + i] = i < n_theta ? theta[(size_t)b * n_theta + i] : 0.0;
    }
    if (threadIdx.x == 0) {
        energy[b] = o[1];
        t[b] = 1.0;
    }
}

// the event that hands the Hessians to the side stream: one per host thread and device, made once
hipEvent_t fork_event()
{
    thread_local hipEvent_t ev[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
    return ev[dev];
}

}  // namespace

// sizeof(oovqe_newton_step_t) as this library was built: a binding checks its mirror of the block against it
extern "C" int oovqe_newton_step_size(void) { return (int)sizeof(oovqe_newton_step_t); }

extern "C" int oovqe_oo_newton_step_batch(const oovqe_newton_step_t* s, oovqe_stream_t stream,
                                          oovqe_stream_t side_stream)
{
    OOVQE_REQUIRE(s, "oo_newton_step_batch: null argument block");
    OOVQE_REQUIRE(s->theta && s->gates && s->g_ao && s->h_ao && s->nuc && s->oao_coeff && s->oao_mo_coeff &&
                      s->mo_coeff && s->kap_row && s->kap_col && s->pairs,
                  "oo_newton_step_batch: null input");
    OOVQE_REQUIRE(s->work_hessian && s->work_eval && s->work_rest && s->out && s->hessian && s->grad && s->energy &&
                      s->flat && s->dp && s->lowest && s->shift && s->info && s->t && s->state && s->flags &&
                      s->points_a && s->points_b && s->trial_oao && s->trial_mo && s->trial_out,
                  "oo_newton_step_batch: null output / workspace");
    const int G = s->batch, nt = s->n_theta, nk = s->n_kappa, n = nt + nk, N = s->N;
    OOVQE_REQUIRE(G >= 1 && G <= 32767 && nt >= 1 && nk >= 1, "oo_newton_step_batch: batch = %d, n_theta = %d, n_kappa = %d",
                  G, nt, nk);
    OOVQE_REQUIRE(n <= oovqe_newton_direction_max_n(), "oo_newton_step_batch: n = %d beyond the direction kernels", n);
    OOVQE_REQUIRE(N <= 48 || s->work_rotate, "oo_newton_step_batch: N = %d needs work_rotate", N);
    hipStream_t st = (hipStream_t)stream, sd = (hipStream_t)side_stream;
    int rc;
    // 1. E, gradient, Hessian
    if ((rc = oovqe_oo_hessian_batch(s->theta, nt, s->gates, s->n_gates, s->n_qubits, s->init_index, s->g_ao, s->h_ao,
                                     s->mo_coeff, s->nuc, N, s->n_occ, s->ncas, s->kap_row, s->kap_col, nk, s->pairs,
                                     s->n_pairs, G, s->work_hessian, s->out, s->hessian, s->eri_flags, s->g_packed,
                                     stream)))
        return rc;
    const long osz1 = (long)oovqe_oo_eval_out_size(nt, nk, s->ncas, 1);
    const long osz0 = (long)oovqe_oo_eval_out_size(nt, nk, s->ncas, 0);
    hipLaunchKernelGGL(step_prepare_kernel, dim3(G), dim3(256), 0, st, s->out, osz1, s->theta, nt, n, s->grad, s->energy,
                       s->flat, s->t);
    OOVQE_CHECK_LAUNCH("oo_newton_step_batch/prepare");
    // 2. directions
    const bool pd = oovqe_newton_direction_has_pd(n, s->aug) != 0 && s->work_pd != nullptr;
    double* info = s->info;
    bool fork_side = false, fork_spec = false;
    if (pd) {
        if ((rc = oovqe_newton_direction_pd(s->hessian, s->grad, n, G, s->lambda_min, s->work_pd, s->dp, s->shift, info,
                                            stream)))
            return rc;
        const bool side_ok = sd != nullptr && sd != st && s->work_rest_side != nullptr;
        const bool spec = s->speculate != 0 && side_ok;
        if (!spec &&
            (rc = oovqe_newton_direction_rest(s->hessian, s->grad, n, G, s->lambda_min, s->mu, s->rho, s->aug, info, 1, 0,
                                              s->work_rest, s->dp, s->lowest, s->shift, stream)))
            return rc;
        if (side_ok) {
            // (the side stream's route is forked BEHIND the trial below: an event record between the direction and
            // the trial's first launch costs that launch ~7 us, and nothing the step's outputs depend on waits for
            // the eigenvalues)
            fork_side = true;
            fork_spec = spec;
        } else if ((rc = oovqe_newton_direction_rest(s->hessian, s->grad, n, G, s->lambda_min, s->mu, s->rho, s->aug, info,
                                                     2, 0, s->work_rest, s->dp, s->lowest, s->shift, stream))) {
            return rc;      // (no side stream: the eigenvalues on the calling stream, behind the others)
        }
    } else {
        OOVQE_CHECK_HIP(hipMemsetAsync(info, 0, (size_t)G * sizeof(double), st), "oo_newton_step_batch: memset");
        if ((rc = oovqe_newton_direction_rest(s->hessian, s->grad, n, G, s->lambda_min, s->mu, s->rho, s->aug, info, 0, 0,
                                              s->work_rest, s->dp, s->lowest, s->shift, stream)))
            return rc;
    }
    // 3. first trial of the line search: t = 1 for every problem
    double* active = s->state;
    double* best = s->state + G;
    double* slope = s->state + 2 * (size_t)G;
    if ((rc = oovqe_linesearch_points(s->flat, s->dp, s->t, s->grad, s->alpha, n, nt, G, s->points_a, s->points_b, slope,
                                      stream)))
        return rc;
    if ((rc = oovqe_rotate_orbitals_batch(s->points_b, s->kap_row, s->kap_col, nk, N, G, s->oao_mo_coeff, s->trial_oao,
                                          nullptr, s->work_rotate, stream)))
        return rc;
    if ((rc = oovqe_matmul_nn_batch(s->oao_coeff, s->trial_oao, N, N, N, G, s->trial_mo, stream))) return rc;
    if ((rc = oovqe_oo_eval_batch(s->points_a, nt, s->gates, s->n_gates, s->n_qubits, s->init_index, s->g_ao, s->h_ao,
                                  s->trial_mo, s->nuc, N, s->n_occ, s->ncas, s->kap_row, s->kap_col, nk, 0, G,
                                  s->work_eval, s->trial_out, s->eri_flags, s->g_packed, stream)))
        return rc;
    if ((rc = oovqe_linesearch_update(s->trial_out + 1, osz0, s->energy, slope, info, s->beta, 1, 0, G, s->t, active,
                                      best, s->flags, stream)))
        return rc;
    if (fork_side) {
        hipEvent_t ev = fork_event();
        OOVQE_REQUIRE(ev, "oo_newton_step_batch: no event for the side stream");
        OOVQE_CHECK_HIP(hipEventRecord(ev, st), "oo_newton_step_batch: hipEventRecord");
        OOVQE_CHECK_HIP(hipStreamWaitEvent(sd, ev, 0), "oo_newton_step_batch: hipStreamWaitEvent");
        // speculate: everything the fast path left (which = 0); else only the eigenvalues of the problems it
        // served (which = 2)
        if ((rc = oovqe_newton_direction_rest(s->hessian, s->grad, n, G, s->lambda_min, s->mu, s->rho, s->aug, info,
                                              fork_spec ? 0 : 2, s->side_wg, s->work_rest_side, s->dp, s->lowest,
                                              s->shift, side_stream)))
            return rc;
    }
    return 0;
}
