// K1: fp64 mode-contraction GEMM on v_mfma_f64_16x16x4_f64 (gfx950).
//
//   INNER:  out[a, j, b] = sum_k Cm[k, j] * T[a, k, b]      T: [A, K, B]  ->  out: [A, J, B]
//   LAST :  out[a, j]    = sum_k T[a, k]  * Cm[k, j]        T: [A, K]     ->  out: [A, J]
//
// This single kernel family carries every dense product of the hot path: the four quarter
// steps of the (pq|rs)->(ij|kl) transform (reference src/auto_oo/oo_energy.py:26-29), the
// one-electron transform C^T h C (oo_energy.py:46), mo_coeff = S^-1/2 C_oao (oo_energy.py:176),
// C U (oo_energy.py:235) and the matrix products inside expm.
//
// MI355X mapping
//   * one wave owns a 16-wide strip of the streamed tensor (16 consecutive b for INNER, 16
//     consecutive rows a for LAST) and ALL J outputs of its j-group (NT <= 13 MFMA tiles):
//     every element of T is fetched from HBM exactly once per j-group, straight into the
//     MFMA operand register (one f64 per lane) - no LDS round trip for the big operand;
//   * the small matrix Cm is shared by all 8 waves of the workgroup: it is staged through LDS
//     in K-chunks of 20 rows, double buffered (global->VGPR prefetch during the MFMAs of the
//     previous chunk), row pitch 16*(NT|1) doubles so that the ds_read_b64 fragment reads of
//     the two 32-lane halves land on disjoint banks;
//   * every global load goes through a buffer descriptor (32-bit lane offset + SGPR base, range
//     check instead of clamps and masks): measured on the N = 200 transform, the address
//     arithmetic, clamps and masks of flat loads cost 8 % of the run time (52 -> 56 TFLOP/s);
//   * f64 MFMA issues one 16x16x4 every 64 cycles per SIMD, so per 13 MFMAs (832 cycles) a wave
//     needs 1 global load + 13 LDS reads: the kernel is MFMA-bound by construction.
#include "common.h"
#include "circuit_small.h"

namespace {

// K rows per LDS chunk = 4 * KS (KS MFMA k-steps): KS = 5 for long contractions, 3 for short ones
// (K is padded with zero rows of Cm to a multiple of the chunk, so a smaller chunk wastes fewer
// MFMAs when K is small, e.g. K = 43 -> 48 instead of 60).
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;

template <int NT, bool LAST, int KSTEPS>
__device__ __forceinline__
void contract_body(const double* __restrict__ T, const double* __restrict__ Cm,
                   double* __restrict__ out, long A, int K, int J, long B, int ldc,
                   long n_items, int nbt, long t_bs, long c_bs, long o_bs, double* lds,
                   const unsigned grid_x)
{
    // blockIdx.z = batch element (independent problems of identical shape)
    T += (long)blockIdx.z * t_bs;
    Cm += (long)blockIdx.z * c_bs;
    out += (long)blockIdx.z * o_bs;
    const int j0 = blockIdx.y * (NT * 16);   // this workgroup's j-group
    constexpr int KC = 4 * KSTEPS;
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int CHUNK = KC * LDJ;
    constexpr int CREG = (CHUNK + NTHREADS - 1) / NTHREADS;
    constexpr int BUF = CREG * NTHREADS;      // chunk buffer, padded so staging stores need no guard
    // lds: [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int nchunks = (K + KC - 1) / KC;
    const long n_groups = (n_items + NWAVES - 1) / NWAVES;

    // One wave = one 16-wide strip of T ("item"); a workgroup walks the item groups
    // blockIdx.x, blockIdx.x + grid_x, ... (persistent), and the chunk pipeline runs ACROSS
    // items: while the last K-chunk of item i is in the MFMA pipe, chunk 0 of item i+1 is already
    // being fetched, so neither the workgroup launch nor the first HBM round trip of an item is
    // ever exposed.
    // T and Cm are read through buffer descriptors: a 32-bit per-lane byte offset (constant for
    // the whole item) plus a wave-uniform base kept in SGPRs, instead of one 64-bit VGPR address
    // per load (the address arithmetic of the flat loads, ~170 VALU instructions per chunk, and
    // their clamps / masks cost a quarter of the MFMA issue slots).  Anything outside the tensor
    // gets an out-of-range offset: the range check drops the load and returns 0.
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const long t_elems = LAST ? A * (long)K : A * (long)K * B;
    struct Strip {
        long tb;        // element index of T[a, 0, 16 bt] (INNER) / T[a, 0] (LAST): wave-uniform
        long a, bcol;
        unsigned tvo;   // this lane's byte offset inside a k-step block, or OOB
        bool active;
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto decode = [&](long group) -> Strip {
        Strip st;
        const long item = group * NWAVES + wave_u;
        st.active = group < n_groups && item < n_items;
        st.a = 0;
        st.bcol = 0;
        st.tb = 0;
        bool tvalid = false;
        if (LAST) {
            st.a = item * 16;
            tvalid = st.active && st.a + lr < A;
            st.tb = st.a * (long)K;
            st.tvo = tvalid ? (unsigned)((lr * (long)K + lq) * sizeof(double)) : OOB;
        } else {
            // 32-bit division (n_items < 2^31 is checked on the host)
            const unsigned ai = (unsigned)item / (unsigned)nbt;
            st.a = ai;
            const long bt = (long)((unsigned)item - ai * (unsigned)nbt);
            st.bcol = bt * 16 + lr;
            tvalid = st.active && st.bcol < B;
            st.tb = st.a * (long)K * B + bt * 16;
            st.tvo = tvalid ? (unsigned)((lq * B + lr) * sizeof(double)) : OOB;
        }
        if (!st.active) st.tb = 0;
        return st;
    };
    const long tstride = LAST ? 1 : B;

    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};

    double creg[CREG];
    double tcur[KSTEPS], tnext[KSTEPS];

    // Staging geometry of Cm, once per workgroup: byte offset of this thread's i-th element inside
    // a chunk (row kk, column j0 + jj), OOB for padding columns / rows; the chunk's first row is
    // added as a scalar offset and rows k >= K fall outside the descriptor (-> 0, the zero padding
    // the MFMAs rely on).
    unsigned cvo[CREG];
#pragma unroll
    for (int i = 0; i < CREG; ++i) {
        const int idx = tid + i * NTHREADS;
        const int kk = idx / LDJ, jj = idx - kk * LDJ;
        const int j = j0 + jj;
        const bool jok = kk < KC && jj < NT * 16 && j < J;
        cvo[i] = jok ? (unsigned)((kk * (long)ldc + j) * sizeof(double)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(Cm), 0, (int)((long)K * ldc * sizeof(double)), 0x00020000);
    auto stage_load = [&](int kbase) {
        const unsigned so = (unsigned)((long)kbase * ldc * sizeof(double));
#pragma unroll
        for (int i = 0; i < CREG; ++i)
            creg[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc_c, cvo[i], so, 0));
    };
    auto stage_store = [&](double* buf) {
#pragma unroll
        for (int i = 0; i < CREG; ++i) buf[tid + i * NTHREADS] = creg[i];
    };
    // k-step s of a chunk: lane (lq, lr) reads T[.., kbase + 4s + lq, ..]; the descriptor is rebased
    // per k-step (T can be larger than the 4 GB a descriptor spans) and ends where T ends
    auto load_t = [&](const Strip& st, int kbase, double* dst) {
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            const long e0 = st.tb + (long)(kbase + 4 * s2) * tstride;   // wave-uniform
            long rem = (t_elems - e0) * (long)sizeof(double);
            rem = rem < 0 ? 0 : (rem > 0xFFFFFFFFL ? 0xFFFFFFFFL : rem);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double*>(T) + e0, 0, (int)(unsigned)rem, 0x00020000);
            dst[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, st.tvo, 0, 0));
        }
    };

    // ---- prologue (once per workgroup) -------------------------------------------------------
    long group = blockIdx.x;
    Strip cur = decode(group);
    stage_load(0);
    load_t(cur, 0, tcur);
    stage_store(lds);
    __syncthreads();
    int par = 0;   // LDS buffer holding the chunk being consumed

    while (group < n_groups) {
        const Strip nxt = decode(group + grid_x);
        for (int c = 0; c < nchunks; ++c) {
            const int kbase = c * KC;
            const bool last = (c + 1 == nchunks);
            // next chunk in the pipeline: chunk c+1 of this item, or chunk 0 of the next item
            const int knext = last ? 0 : kbase + KC;
            const Strip& stn = last ? nxt : cur;
            // One straight-line code path for every chunk (rows k >= K of the staged Cm are zero,
            // so the padded k-steps of the last chunk add nothing).  The Cm fragments of k-step
            // s+1 are read from LDS while the MFMAs of k-step s issue (two-stage pipeline).  The
            // global prefetch (address arithmetic + loads) sits in the SAME scheduling region as
            // the MFMAs of k-step 0, and the LDS staging stores in the region of the last k-step,
            // so their VALU work fills the 64-cycle MFMA issue gaps instead of running ahead of
            // the first MFMA after every barrier.
            const double* buf = lds + par * BUF + lq * LDJ + lr;
            double cv[2][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) cv[0][t] = buf[t * 16];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if (s + 1 < KSTEPS) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) cv[(s + 1) & 1][t] = buf[(s + 1) * 4 * LDJ + t * 16];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = LAST ? mfma_f64(tcur[s], cv[s & 1][t], acc[t])
                                  : mfma_f64(cv[s & 1][t], tcur[s], acc[t]);
                if (s == 0) {
                    stage_load(knext);
                    load_t(stn, knext, tnext);
                }
                if (s == KSTEPS - 1) stage_store(lds + (par ^ 1) * BUF);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            par ^= 1;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) tcur[s] = tnext[s];
        }
        // ---- epilogue of this item (stores drain while the next item's MFMAs run) -------------
        // one running pointer + a scheduling fence per tile: otherwise all 4*NT 64-bit store
        // addresses are materialised at once (2 VGPRs each) on top of the accumulators
        if (cur.active) {
            if (LAST) {
                double* op = out + (cur.a + lq) * (long)J + j0 + lr;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = j0 + t * 16 + lr;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const long row = cur.a + lq + 4 * i;
                        if (row < A && col < J) op[(long)(4 * i) * J + t * 16] = acc[t][i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (cur.bcol < B) {
                double* op = out + (cur.a * (long)J + j0 + lq) * B + cur.bcol;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = j0 + t * 16 + lq + 4 * i;
                        if (j < J) op[(long)(t * 16 + 4 * i) * B] = acc[t][i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        cur = nxt;
        group += grid_x;
    }
}

template <int NT, bool LAST, int KSTEPS>
__global__ __launch_bounds__(NTHREADS, 2)
void contract_kernel(const double* __restrict__ T, const double* __restrict__ Cm,
                     double* __restrict__ out, long A, int K, int J, long B, int ldc,
                     long n_items, int nbt, long t_bs, long c_bs, long o_bs)
{
    extern __shared__ double lds[];
    contract_body<NT, LAST, KSTEPS>(T, Cm, out, A, K, J, B, ldc, n_items, nbt, t_bs, c_bs, o_bs, lds,
                                    gridDim.x);
}

// The same contraction with one extra workgroup column (blockIdx.x == gridDim.x - 1) that runs
// independent small-circuit evaluations (circuit_small.h) instead: instance b = blockIdx.z of the
// batch, on the workgroups with blockIdx.y == 0.  Used by the OO evaluation, whose circuit + RDM
// step and p -> n contraction are both short, mutually independent launches.
template <int NT, bool LAST, int KSTEPS>
__global__ __launch_bounds__(NTHREADS, 2)
void contract_circuit_kernel(const double* __restrict__ T, const double* __restrict__ Cm,
                             double* __restrict__ out, long A, int K, int J, long B, int ldc,
                             long n_items, int nbt, long t_bs, long c_bs, long o_bs,
                             oovqe_circuit_job_t cj)
{
    extern __shared__ double lds[];
    static_assert(SMALL_THREADS == NTHREADS, "the circuit body needs the K1 workgroup size");
    if (blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0 && (int)blockIdx.z < cj.count)
            circuit_rdm_small_body(cj.theta, cj.n_theta, cj.gates, cj.n_gates, cj.n_qubits, cj.ncas,
                                   cj.init_index, cj.n_tan, nullptr, nullptr, cj.gamma, cj.Gamma,
                                   (int)blockIdx.z, lds);
        return;
    }
    contract_body<NT, LAST, KSTEPS>(T, Cm, out, A, K, J, B, ldc, n_items, nbt, t_bs, c_bs, o_bs, lds,
                                    gridDim.x - 1);
}

template <int NT, bool LAST, int KS>
int launch_nt(const double* T, const double* Cm, double* out, long A, int K, int J, long B,
              int ldc, int ngroups, long n_items, int nbt, int batch, long t_bs, long c_bs, long o_bs,
              hipStream_t st, const oovqe_circuit_job_t* cj = nullptr)
{
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int KC = 4 * KS;
    constexpr int CREG = (KC * LDJ + NTHREADS - 1) / NTHREADS;
    size_t lds_bytes = (size_t)2 * CREG * NTHREADS * sizeof(double);
    // persistent grid: at most ~2 workgroups per CU in total, each walks many item groups
    const long ngroups_items = (n_items + NWAVES - 1) / NWAVES;
    long per_slice = 512 / ((long)ngroups * batch);
    if (per_slice < 1) per_slice = 1;
    const long nblocks = ngroups_items < per_slice ? ngroups_items : per_slice;
    if constexpr (KS == 12) {
        if (cj) {
            if (cj->lds_bytes > lds_bytes) lds_bytes = cj->lds_bytes;
            static size_t attr_bytes = 0;   // per instantiation
            if (lds_bytes > attr_bytes) {
                hipError_t e = hipFuncSetAttribute((const void*)contract_circuit_kernel<NT, LAST, KS>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds_bytes);
                if (e != hipSuccess) {
                    oovqe_set_error("mode_contract: hipFuncSetAttribute: %s", hipGetErrorString(e));
                    return OOVQE_ERR_HIP;
                }
                attr_bytes = lds_bytes;
            }
            hipLaunchKernelGGL((contract_circuit_kernel<NT, LAST, KS>),
                               dim3((unsigned)nblocks + 1, (unsigned)ngroups, (unsigned)batch),
                               dim3(NTHREADS), lds_bytes, st, T, Cm, out, A, K, J, B, ldc, n_items, nbt,
                               t_bs, c_bs, o_bs, *cj);
            OOVQE_CHECK_LAUNCH("mode_contract+circuit");
            return 0;
        }
    } else {
        OOVQE_REQUIRE(!cj, "mode_contract: this shape cannot host circuit workgroups");
    }
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)contract_kernel<NT, LAST, KS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) {
            oovqe_set_error("mode_contract: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((contract_kernel<NT, LAST, KS>),
                       dim3((unsigned)nblocks, (unsigned)ngroups, (unsigned)batch), dim3(NTHREADS),
                       lds_bytes, st, T, Cm, out, A, K, J, B, ldc, n_items, nbt, t_bs, c_bs, o_bs);
    OOVQE_CHECK_LAUNCH("mode_contract");
    return 0;
}

template <bool LAST, int KS>
int launch_group(int nt, const double* T, const double* Cm, double* out, long A, int K, int J,
                 long B, int ldc, int ngroups, long n_items, int nbt, int batch, long t_bs, long c_bs,
                 long o_bs, hipStream_t st)
{
    switch (nt) {
#define OOVQE_CASE(n) \
    case n:                                                                                    \
        return launch_nt<n, LAST, KS>(T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, t_bs, \
                                      c_bs, o_bs, st);
        OOVQE_CASE(1) OOVQE_CASE(2) OOVQE_CASE(3) OOVQE_CASE(4) OOVQE_CASE(5) OOVQE_CASE(6)
        OOVQE_CASE(7) OOVQE_CASE(8) OOVQE_CASE(9) OOVQE_CASE(10) OOVQE_CASE(11) OOVQE_CASE(12)
        OOVQE_CASE(13)
#undef OOVQE_CASE
    }
    oovqe_set_error("mode_contract: bad tile count %d", nt);
    return OOVQE_ERR_ARG;
}

// Short contractions (K <= 48) with few tiles: the whole small matrix is ONE staged chunk
// (KSTEPS = 12), so an item costs one barrier instead of four chunk rounds -- these launches are
// latency-bound (p -> n step of an evaluation: 43 x 43 x 729 per geometry).
template <bool LAST>
int launch_short(int nt, const double* T, const double* Cm, double* out, long A, int K, int J, long B,
                 int ldc, int ngroups, long n_items, int nbt, int batch, long t_bs, long c_bs, long o_bs,
                 hipStream_t st, const oovqe_circuit_job_t* cj)
{
    switch (nt) {
#define OOVQE_CASE(n) \
    case n:                                                                                    \
        return launch_nt<n, LAST, 12>(T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, t_bs, \
                                      c_bs, o_bs, st, cj);
        OOVQE_CASE(1) OOVQE_CASE(2) OOVQE_CASE(3) OOVQE_CASE(4)
#undef OOVQE_CASE
    }
    oovqe_set_error("mode_contract: bad tile count %d", nt);
    return OOVQE_ERR_ARG;
}

}  // namespace

int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st);

int oovqe_mode_contract_impl(const double* T, const double* Cm, double* out, long A, int K, int J,
                             long B, int ldc, int last, hipStream_t st)
{
    return oovqe_mode_contract_batched(T, Cm, out, A, K, J, B, ldc, last, 1, 0, 0, 0, st);
}

// Can the launch for this shape host circuit workgroups (single-chunk kernels, K <= 48, <= 4 tiles)?
// Mirrors the dispatch of oovqe_mode_contract_batched_circ below.
static int contract_plan(long A, int K, int J, long B, int last, int batch, int* nt_out, int* ngroups_out,
                         long* n_items_out, int* nbt_out)
{
    const int JT = (J + 15) / 16;
    long n_items;
    int nbt = 1;
    if (last) {
        n_items = (A + 15) / 16;
    } else {
        const long nb = (B + 15) / 16;
        nbt = (int)nb;
        n_items = A * nb;
    }
    int nt = JT < 13 ? JT : 13;
    const long wgs = (n_items + NWAVES - 1) / NWAVES;
    // (never past one resident round of workgroups, ~2 per CU: a second round doubles the latency of
    // these short launches -- 17 -> 9 us for the p -> n step of a 64-geometry evaluation)
    while (nt > 1 && wgs * ((JT + nt - 1) / nt) * batch < 512) {
        const int nt2 = (nt + 1) / 2;
        if (wgs * ((JT + nt2 - 1) / nt2) * batch > 512) break;
        nt = nt2;
    }
    const int ngroups = (JT + nt - 1) / nt;
    nt = (JT + ngroups - 1) / ngroups;   // even split
    *nt_out = nt;
    *ngroups_out = ngroups;
    *n_items_out = n_items;
    *nbt_out = nbt;
    return 0;
}

int oovqe_contract_hosts_circuit(long A, int K, int J, long B, int last, int batch)
{
    int nt, ngroups, nbt;
    long n_items;
    contract_plan(A, K, J, B, last, batch, &nt, &ngroups, &n_items, &nbt);
    return K <= 48 && nt <= 4;
}

int oovqe_mode_contract_batched_circ(const double* T, const double* Cm, double* out, long A, int K, int J,
                                     long B, int ldc, int last, int batch, long t_bs, long c_bs,
                                     long o_bs, hipStream_t st, const oovqe_circuit_job_t* cj);

int oovqe_mode_contract_batched(const double* T, const double* Cm, double* out, long A, int K, int J,
                                long B, int ldc, int last, int batch, long t_bs, long c_bs, long o_bs,
                                hipStream_t st)
{
    return oovqe_mode_contract_batched_circ(T, Cm, out, A, K, J, B, ldc, last, batch, t_bs, c_bs, o_bs, st,
                                            nullptr);
}

int oovqe_mode_contract_batched_circ(const double* T, const double* Cm, double* out, long A, int K, int J,
                                     long B, int ldc, int last, int batch, long t_bs, long c_bs,
                                     long o_bs, hipStream_t st, const oovqe_circuit_job_t* cj)
{
    OOVQE_REQUIRE(batch >= 1 && batch <= 65535, "mode_contract: batch=%d", batch);
    OOVQE_REQUIRE((double)A * (double)((B + 15) / 16) < 2.0e9, "mode_contract: too many strips");
    OOVQE_REQUIRE(T && Cm && out, "mode_contract: null pointer");
    OOVQE_REQUIRE(A >= 1 && K >= 1 && J >= 1 && B >= 1 && ldc >= J,
                  "mode_contract: bad dims A=%ld K=%d J=%d B=%ld ldc=%d", A, K, J, B, ldc);
    OOVQE_REQUIRE(!last || B == 1, "mode_contract: last-mode needs B == 1");
    OOVQE_REQUIRE(last || (B + 15) / 16 <= 0x7fffffffL, "mode_contract: B too large");
    // tiles per wave: as many as fit (T is then streamed once), fewer when the problem is too
    // small to fill 256 CUs with 8-wave workgroups.  The j-groups are grid.y of ONE launch.
    int nt, ngroups, nbt;
    long n_items;
    contract_plan(A, K, J, B, last, batch, &nt, &ngroups, &n_items, &nbt);
    OOVQE_REQUIRE(ngroups <= 65535, "mode_contract: J too large");
    // chunk depth: 20 rows when that wastes <= 5 % of the MFMAs on zero padding, else 12 rows
    const int pad20 = ((K + 19) / 20) * 20, pad12 = ((K + 11) / 12) * 12;
    const bool deep = pad20 <= pad12 || pad20 * 100 <= K * 105;
    int rc;
    OOVQE_REQUIRE(!cj || (K <= 48 && nt <= 4), "mode_contract: this shape cannot host circuit workgroups");
    if (K <= 48 && nt <= 4)
        rc = last ? launch_short<true>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, t_bs,
                                       c_bs, o_bs, st, cj)
                  : launch_short<false>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, t_bs,
                                        c_bs, o_bs, st, cj);
    else if (deep)
        rc = last ? launch_group<true, 5>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch,
                                     t_bs, c_bs, o_bs, st)
                  : launch_group<false, 5>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch,
                                     t_bs, c_bs, o_bs, st);
    else
        rc = last ? launch_group<true, 3>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch,
                                     t_bs, c_bs, o_bs, st)
                  : launch_group<false, 3>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch,
                                     t_bs, c_bs, o_bs, st);
    if (rc) return rc;
    return 0;
}

extern "C" int oovqe_mode_contract(const double* T, const double* Cm, double* out, int64_t A, int K,
                                   int J, int64_t B, int ldc, int last, oovqe_stream_t stream)
{
    return oovqe_mode_contract_impl(T, Cm, out, (long)A, K, J, (long)B, ldc, last,
                                    (hipStream_t)stream);
}

extern "C" int oovqe_matmul_nn(const double* A, const double* B, int M, int K, int N, double* out,
                               oovqe_stream_t stream)
{
    // out[m,n] = sum_k A[m,k] B[k,n]  == LAST with T = A ([M,K]), Cm = B ([K,N])
    return oovqe_mode_contract_impl(A, B, out, M, K, N, 1, N, 1, (hipStream_t)stream);
}

extern "C" int oovqe_matmul_tn(const double* A, const double* B, int M, int K, int N, double* out,
                               oovqe_stream_t stream)
{
    // out[m,n] = sum_k A[k,m] B[k,n]  == INNER with a single slab: Cm = A ([K,M]), T = B ([1,K,N])
    return oovqe_mode_contract_impl(B, A, out, 1, K, M, N, M, 0, (hipStream_t)stream);
}

extern "C" int oovqe_general_4index_transform(const double* M, const double* C0, const double* C1,
                                              const double* C2, const double* C3, int N, double* out,
                                              double* work, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(M && C0 && C1 && C2 && C3 && out && work, "4index_transform: null pointer");
    OOVQE_REQUIRE(N >= 1, "4index_transform: N=%d", N);
    OOVQE_REQUIRE(out != M && out != work && work != M, "4index_transform: aliased buffers");
    hipStream_t st = (hipStream_t)stream;
    const long n = N, n2 = n * n, n3 = n2 * n;
    int rc;
    // 'pi,pqrs->iqrs'
    if ((rc = oovqe_mode_contract_impl(M, C0, work, 1, N, N, n3, N, 0, st))) return rc;
    // 'qj,iqrs->ijrs'
    if ((rc = oovqe_mode_contract_impl(work, C1, out, n, N, N, n2, N, 0, st))) return rc;
    // 'rk,ijrs->ijks'
    if ((rc = oovqe_mode_contract_impl(out, C2, work, n2, N, N, n, N, 0, st))) return rc;
    // 'sl,ijks->ijkl'
    if ((rc = oovqe_mode_contract_impl(work, C3, out, n3, N, N, 1, N, 1, st))) return rc;
    return 0;
}
