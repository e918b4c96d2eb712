// Shared helpers for liboovqe_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/oovqe.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
// two adjacent doubles fetched with one 16-byte load from an address that is only 8-byte aligned
// (gfx950 global loads need dword alignment only)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

#define OOVQE_ERR_ARG   (-1)
#define OOVQE_ERR_HIP   (-2)
#define OOVQE_ERR_SIZE  (-3)

void oovqe_set_error(const char* fmt, ...);
void oovqe_profile_mark_start(hipStream_t st);
void oovqe_profile_mark_start_l(hipStream_t st, int label);
void oovqe_profile_mark_stop(hipStream_t st);

// Test / measurement switches between realisations that compute the same numbers (set through
// oovqe_debug_set_option, never from the environment: the production path does not depend on the
// caller's environment variables).  All default to 0.
enum oovqe_option_t {
    OOVQE_OPT_HALF_STREAM_OLD = 0,   // pre-persistent streaming kernel for N > 48
    OOVQE_OPT_GM_TWO_PER_CU,         // force the 128-VGPR build of sym_gm_kernel
    OOVQE_OPT_GM_ONE_PER_CU,         // force its one-workgroup-per-CU build
    OOVQE_OPT_FUSED_CHUNKS,          // n > 0: force the batched plan with n chunks of q
    OOVQE_OPT_TRI_PLAIN_W,           // W = CUs / batch in half_tri_kernel (no cost model)
    OOVQE_OPT_CAS_UNFUSED,           // T2 path
    OOVQE_OPT_SYM_NO_RS,             // ignore the r<->s flag
    OOVQE_OPT_SYM_MIRROR,            // mirrored T2 instead of the packed triangle
    OOVQE_OPT_SYM_SIMPLE,            // one-slab-per-wave triangle kernel
    OOVQE_OPT_SYM_TWO_STEP,          // q->x kernel + K1 instead of the one-launch kernel
    OOVQE_OPT_NO_RIDE,               // 1: circuit + RDM step as its own launch whatever the batch; 2: riding whatever the batch
    OOVQE_OPT_TRI_MODE,              // packed-triangle stage 1: 1 operand loads, 2 LDS-DMA, 3/4 contiguous loads
    OOVQE_OPT_K1_NO_PAIR,            // K1: never the two-strips-per-wave kernel
    OOVQE_OPT_K1_FORCE_WIDE,         // K1: INNER contractions on the long-stride kernels (one descriptor per k-step)
    OOVQE_OPT_GM_PLAIN_GRID,         // sym_gm_kernel: (tile, geometry) grid instead of the XCD-aware 1-D one
    OOVQE_OPT_NEWTON_ONE_WG,         // oovqe_newton_direction: the one-workgroup tridiagonalisation kernel (n <= 480)
    OOVQE_OPT_SECTOR_UNFUSED,        // sector RDMs / adjoint: E_pq vectors through memory (the round-2 kernels)
    OOVQE_OPT_SECTOR_PROBE,          // timing probe of sector_rdm_fused_kernel (wrong results): 1 no chunk build, 2 no MFMA
    OOVQE_OPT_HESS_VK_PASS,          // orbital Hessian: the K-type quarter transform as its own pass over the AO tensor (round 2)
    OOVQE_OPT_HESS_OWN_STAGE1,       // Hessian call: the evaluation streams the integrals itself instead of taking J from the orbital Hessian's T2
    OOVQE_OPT_PANEL_ROWS,            // cas_panel_kernel: general indices per workgroup (0: chosen by the host code)
    OOVQE_OPT_K1_FORCE_NT,           // K1 / K1P: this many 16-wide tiles of J per wave (0: all that fit, up to 13)
    OOVQE_OPT_NEWTON_NO_CHOL,        // oovqe_newton_direction: never the Cholesky fast path (band route for every problem)
    OOVQE_OPT_TILES_VARIANT,         // half_tiles_kernel (measurement, M in 17..32): 1 / 2 = ring of 4 / 6 tiles instead of 8, 4 = default cache policy for the loads
    OOVQE_OPT_SECTOR_LAMBDA_W,       // sector adjoint: 1 = lambda through W = Ms^T V in memory (round 3) whatever the batch, 2 = the string-driven form whatever the batch, 3 = the same (kept for the tools), 4 = that with multiplier / helper waves (sector_lambda_pipe_kernel: measured, slower) (0: by batch size)
    OOVQE_OPT_SECTOR_RDM_R3,         // sector RDMs: 1 = the round-3 fused kernel (chunks of 128 consecutive determinants) whatever the batch, 2 = row chunks in the sigma basis whatever the batch (0: by batch size)
    OOVQE_OPT_GM_THREE_PER_CU,       // sym_gm_kernel at N = 41 ... 44: 1 = the three-workgroups-per-CU build (measured: slower)
    OOVQE_OPT_PANEL_NO_W,            // cas_panel_kernel always stages h_ao and forms its rows of C^T h itself (no W from the circuit launch)
    OOVQE_OPT_STAGE1_FREE_RUN,       // 1: N^4 sweeps enqueued on different streams are not ordered one after the other
    OOVQE_OPT_ONE_STREAM,            // 1: every launch of a call on the caller's stream (no chain of a call on the library's internal streams)
    OOVQE_OPT_COUNT
};
int oovqe_opt(int id);
// raise a kernel's dynamic-LDS limit to `bytes` (cached per kernel AND device, thread-safe); 0 or OOVQE_ERR_HIP
int oovqe_ensure_dynamic_lds(const void* kernel, size_t bytes);
void oovqe_note_stage1(const char* fmt, ...);
// the next event of a small per-device ring (timing disabled; nullptr on failure), and the library's internal
// stream of the current device (k = 0; made on first use, non-blocking; nullptr on failure or with option
// one_stream): independent chains of ONE library call run beside each other, forked from and joined to the
// caller's stream inside the call -- the caller sees in-order stream semantics.  ONE stream: HIP multiplexes a
// process' streams over four hardware queues; the caller's stream and the two side streams of the Python layer
// (ops.side_streams) take three of them, and streams that share a queue run one after the other (measured: with
// two internal streams the deferred evaluations of OO_pqc_batch lost all their overlap).
hipEvent_t oovqe_internal_event();
hipStream_t oovqe_internal_stream(int k);
// Around every launch of the packed N^4 sweep: sweeps enqueued on DIFFERENT streams (calls in flight on two streams:
// OO_pqc_batch.evaluate_deferred) run one after the other -- each is sized to fill the chip, and the HIP events that
// bracket a launch for the roofline then time the sweep, not its wait for the CUs of the previous one.
int oovqe_stage1_enter(hipStream_t st);
int oovqe_stage1_leave(hipStream_t st);

#define OOVQE_CHECK_LAUNCH(name)                                                          \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) {                                                          \
            oovqe_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return OOVQE_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define OOVQE_CHECK_HIP(call, name)                                                        \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            oovqe_set_error("%s: %s", name, hipGetErrorString(e__));                      \
            return OOVQE_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define OOVQE_REQUIRE(cond, ...)                                                          \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            oovqe_set_error(__VA_ARGS__);                                                 \
            return OOVQE_ERR_ARG;                                                         \
        }                                                                                 \
    } while (0)

// Small-circuit evaluations riding along a K1 launch (contract.hip: contract_circuit_kernel):
// `count` independent instances (one per geometry of a batch), see circuit_small.h.
struct oovqe_circuit_job_t {
    const double* theta;         // [count][n_theta]
    const oovqe_gate_t* gates;
    double* gamma;               // [count][1 + n_tan][a^2]
    double* Gamma;               // [count][1 + n_tan][a^4]
    int n_theta, n_gates, n_qubits, ncas, n_tan, count;
    uint32_t init_index;
    size_t lds_bytes;            // oovqe_small_circuit_lds_bytes(...)
};

// v_mfma_f64_16x16x4_f64: D[16x16] += A[16x4] B[4x16].
// lane l supplies A[m = l&15][k = l>>4] and B[k = l>>4][n = l&15]; receives
// D[m = (l>>4) + 4*i][n = l&15] in element i of the accumulator (cdna_hip_programming.md:161).
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}


// Byte count for a buffer descriptor: rem clamped to [0, 2^32 - 1] with 32-bit scalar compares (there
// is no 64-bit scalar compare: the plain form runs on the VALU, and VALU instructions cost MFMA time).
__device__ __forceinline__ long clamp_u32(long rem)
{
    int hi = (int)(rem >> 32);
    const unsigned lo = (unsigned)rem;
    asm("" : "+s"(hi));            // (keeps the optimiser from folding this back into a 64-bit compare)
    return (long)(hi < 0 ? 0u : (hi != 0 ? 0xFFFFFFFFu : lo));
}
