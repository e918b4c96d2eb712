// K1P: the INNER mode contraction of contract.hip with TWO 16-wide strips of T per wave.
//
//   out[a, j, b] = sum_k Cm[k, j] * T[a, k, b]        T: [A, K, B]  ->  out: [A, J, B],  B even
//
// (Reference: the quarter steps 'pi,pqrs->iqrs', 'qj,iqrs->ijrs', 'rk,ijrs->ijks' of
// src/auto_oo/oo_energy.py:26-28.)
//
// What the measurements on contract.hip's kernel said (tools/k1_standalone.hip, N = 200): per 65
// MFMAs a wave issues 35 LDS fragment reads and 5 loads of T, and per strip 52 result stores; the
// stores alone cost 10 % (a store instruction takes the SIMD ~70 cycles whatever its width, and
// its cost grows with the number of separate rows it touches), the T loads 7 %.  Here a wave owns
// 32 adjacent columns b: lane (lq, lr) loads the PAIR T[k, 2 lr], T[k, 2 lr + 1] with one 16-byte
// load; the even columns are the B operand of one product, the odd columns of a second one, and
// both use the same Cm fragment.  Per MFMA that is half the fragment reads and half the load
// instructions, and accumulator i of the two products in a lane are adjacent elements of out: one
// 16-byte store writes 4 rows x 256 contiguous bytes, half the instructions of before at the same
// number of rows.  The 2 x NT accumulator tiles (208 registers at NT = 13) live in AGPRs: four
// waves per workgroup, one per SIMD, 512 registers each.
//
// Everything else follows contract.hip: Cm staged through LDS in K-chunks (double buffered), one
// buffer descriptor per chunk with the k-step as scalar offset, fragment reads and prefetch parts
// between groups of MFMAs, the last chunk of a strip tile-outer with its stores between the MFMAs.
#include "common.h"

// cache policy of the result stores / the T loads (buffer aux bits; 2 = nt: streaming)
#ifndef K1_AUX_ST
#define K1_AUX_ST 0
#endif
#ifndef K1_AUX_LD
#define K1_AUX_LD 0
#endif

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#if defined(OOVQE_K1_PROBE) && (OOVQE_K1_PROBE & 64)
// tools/k1_standalone.hip: clock marks of wave 0 of workgroup 0 from its 4th strip on
__device__ long long g_k1p_marks[128];
__device__ long long g_k1p_clk[2];     // workgroup 0: core cycles and 100 MHz ticks of the whole kernel
#define K1P_CLK_START const long long k1p_c0 = clock64(), k1p_w0 = wall_clock64();
#define K1P_CLK_END                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                              \
        g_k1p_clk[0] = clock64() - k1p_c0;                                                     \
        g_k1p_clk[1] = wall_clock64() - k1p_w0;                                                \
    }
#define K1P_MARK(code)                                                                         \
    do {                                                                                       \
        if (k1_items >= 3 && k1_m < 64 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
            g_k1p_marks[2 * k1_m] = (code);                                                    \
            g_k1p_marks[2 * k1_m + 1] = clock64();                                             \
        }                                                                                      \
        if (k1_items >= 3) ++k1_m;                                                             \
    } while (0)
#else
#define K1P_MARK(code) do { } while (0)
#define K1P_CLK_START
#define K1P_CLK_END
#endif

namespace {

constexpr int PWAVES = 4;
constexpr int PTHREADS = PWAVES * 64;

// (Round 4 tried narrow builds, NT <= 7 = two j-groups at J = 200, with TWO workgroups per CU -- 190
// registers, the waves of a SIMD then belong to different workgroups and are not in step: 55.9 TFLOP/s against
// 62.4 with NT = 13 and one workgroup per CU at N = 200.  T is streamed once per j-group and every MFMA gets
// half the fragment reuse: occupancy does not pay for that.  tools/k1_standalone.hip 200 nt=7.)
template <int NT, int KSTEPS>
__global__ __launch_bounds__(PTHREADS, 1)
void contract_pair_kernel(const double* __restrict__ T, const double* __restrict__ Cm,
                          double* __restrict__ out, long A, int K, int J, long B, int ldc,
                          long n_items, int nbt, long t_bs, long c_bs, long o_bs)
{
    extern __shared__ double lds[];
    T += (long)blockIdx.z * t_bs;
    Cm += (long)blockIdx.z * c_bs;
    out += (long)blockIdx.z * o_bs;
    const int j0 = blockIdx.y * (NT * 16);
    constexpr int NW = PWAVES, NTH = PTHREADS;
    constexpr int KC = 4 * KSTEPS;
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int CHUNK = KC * LDJ;
    constexpr int CREG = (CHUNK + NTH - 1) / NTH;
    constexpr int BUF = CREG * NTH;
    constexpr unsigned OOB = 0xFFFFFFFFu;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int nchunks = (K + KC - 1) / KC;
    const long n_groups = (n_items + NW - 1) / NW;
    const unsigned grid_x = gridDim.x;

    struct Strip {
        long tb;        // element index of T[a, 0, 32 bt]: wave-uniform
        long te;        // element index where slab a of T ends (k-steps past K read nothing)
        long ob, oe;    // element indices in out: the strip's first result, the end of its slab
        unsigned vo;    // this lane's byte offset (row lq, columns 2 lr, 2 lr + 1), or OOB
        bool active;
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto decode = [&](long group) -> Strip {
        Strip st;
        const long item = group * NW + wave_u;
        st.active = group < n_groups && item < n_items;
        const unsigned ai = (unsigned)item / (unsigned)nbt;   // (n_items < 2^31 is checked on the host)
        const long bt = (long)((unsigned)item - ai * (unsigned)nbt);
        const bool valid = st.active && bt * 32 + 2 * lr < B;
        st.tb = (long)ai * K * B + bt * 32;
        st.te = ((long)ai + 1) * K * B;
        st.ob = ((long)ai * J + j0) * B + bt * 32;
        st.oe = ((long)ai * J + J) * B;
        st.vo = valid ? (unsigned)((lq * B + 2 * lr) * sizeof(double)) : OOB;
        if (!st.active) st.tb = 0, st.te = 0, st.ob = 0, st.oe = 0;
        return st;
    };

    K1P_CLK_START
    d4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[0][t] = acc[1][t] = d4{0.0, 0.0, 0.0, 0.0};
    double creg[CREG];
    d2 tcur[KSTEPS], tnext[KSTEPS];

    // staging geometry of Cm (see contract.hip)
    unsigned cvo[CREG];
#pragma unroll
    for (int i = 0; i < CREG; ++i) {
        const int idx = tid + i * NTH;
        const int kk = idx / LDJ, jj = idx - kk * LDJ;
        const int j = j0 + jj;
        const bool jok = kk < KC && jj < NT * 16 && j < J;
        cvo[i] = jok ? (unsigned)((kk * (long)ldc + j) * sizeof(double)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(Cm), 0, (int)((long)K * ldc * sizeof(double)), 0x00020000);
    const unsigned step_bytes = (unsigned)(4 * B * (long)sizeof(double));

    constexpr int NPF = CREG + KSTEPS;   // prefetch items: CREG loads of Cm, KSTEPS loads of T
    int pf_kn = 0;
    unsigned pf_cso = 0, pf_rem = 0;
    __amdgpu_buffer_rsrc_t pf_tr = rsrc_c;
    // start of a chunk's prefetch: the k offset of the Cm rows and the descriptor of the T rows
    auto prefetch_begin = [&](const Strip& stn, int knext) {
        pf_kn = knext;
        asm volatile("" : "+s"(pf_kn));   // keeps the SALU arithmetic behind the MFMAs it follows
        pf_cso = (unsigned)((long)pf_kn * ldc * sizeof(double));
        const long e1 = stn.tb + (long)pf_kn * B;   // wave-uniform
        long rem = (stn.te - e1) * (long)sizeof(double);
        rem = clamp_u32(rem);
        pf_rem = (unsigned)rem;
        pf_tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T) + e1, 0, (int)pf_rem, 0x00020000);
    };
    auto prefetch_c_part = [&](int part, int np) {
        const int lo = part * CREG / np, hi = (part + 1) * CREG / np;
#pragma unroll
        for (int q = 0; q < CREG; ++q)
            if (q >= lo && q < hi)
                creg[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc_c, cvo[q], pf_cso, 0));
    };
    auto prefetch_t = [&](const Strip& stn, int s2, d2& dst) {
        unsigned so = (unsigned)s2 * step_bytes;
        so = so < pf_rem ? so : pf_rem;
        dst = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(pf_tr, stn.vo, so, K1_AUX_LD));
    };
    // Cm and T together in np parts (prologue and last chunk of a strip: T goes to tnext)
    auto prefetch_part = [&](const Strip& stn, int knext, int part, int np) {
        const int lo = part * NPF / np, hi = (part + 1) * NPF / np;
        if (part == 0) prefetch_begin(stn, knext);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            if (q < lo || q >= hi) continue;
            if (q < CREG) {
                creg[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc_c, cvo[q], pf_cso, 0));
            } else {
                prefetch_t(stn, q - CREG, tnext[q - CREG]);
            }
        }
    };
    auto stage_store_part = [&](double* buf, int part, int np) {
        const int lo = part * CREG / np, hi = (part + 1) * CREG / np;
#pragma unroll
        for (int i = 0; i < CREG; ++i)
            if (i >= lo && i < hi) buf[tid + i * NTH] = creg[i];
    };
    int par = 0;
    auto flip = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        par ^= 1;
    };
    // (last chunk of a strip only: its T prefetch cannot land in registers the chunk reads to its end.
    // A VALU move costs MFMA time on gfx950, tools/mfma_dep_probe.hip: the other chunks reload tcur[s]
    // in place as soon as k-step s is done.)
    auto rotate = [&]() {
        flip();
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            tcur[s] = tnext[s];
            asm volatile("" : "+v"(tcur[s]));   // the wait for these loads happens here (contract.hip)
        }
    };
    // tile t of both products: accumulator i of the even and of the odd columns is one 16-byte pair
    auto store_tile = [&](const Strip& st, int t) {
        const long e0 = st.ob + (long)t * 16 * B;   // wave-uniform
        long rem = (st.oe - e0) * (long)sizeof(double);
        rem = clamp_u32(rem);
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(out + e0, 0, (int)(unsigned)rem, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double v0 = acc[0][t][i], v1 = acc[1][t][i];
            const d2 v = {v0, v1};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, st.vo, (unsigned)i * step_bytes, K1_AUX_ST);
        }
    };

    // ---- prologue ------------------------------------------------------------------------------
    long group = blockIdx.x;
    Strip cur = decode(group);
    prefetch_part(cur, 0, 0, 1);
    stage_store_part(lds, 0, 1);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        tcur[s] = tnext[s];
        asm volatile("" : "+v"(tcur[s]));
    }

    [[maybe_unused]] int k1_items = 0, k1_m = 0;
    while (group < n_groups) {
        const Strip nxt = decode(group + grid_x);
        // ---- all chunks but the last: k-step outer, tile inner -----------------------------------
        // The next chunk's T operand of k-step s is loaded over tcur[s] once k-step s is done; the last
        // NDB k-steps, whose loads would be issued too late for the next chunk's start (the last chunk
        // of a strip runs tile-outer and needs every k-step at its first tile), go through tnext and
        // are copied at the barrier.
        constexpr int NDB = KSTEPS >= 4 ? 2 : 1;
        for (int c = 0; c + 1 < nchunks; ++c) {
            const int knext = (c + 1) * KC;
            K1P_MARK(c);
            const double* buf = lds + par * BUF + lq * LDJ + lr;
            double cv[2][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) cv[0][t] = buf[t * 16];
            constexpr int NP = (NT + 1) / 2;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 2 * p2; t < 2 * p2 + 2 && t < NT; ++t) {
                        acc[0][t] = mfma_f64(cv[s & 1][t], tcur[s].x, acc[0][t]);
                        acc[1][t] = mfma_f64(cv[s & 1][t], tcur[s].y, acc[1][t]);
                    }
#pragma unroll
                    for (int t = 2 * p2; t < 2 * p2 + 2 && t < NT; ++t)
                        if (s + 1 < KSTEPS) cv[(s + 1) & 1][t] = buf[(s + 1) * 4 * LDJ + t * 16];
                    if (s == 0 && p2 == 0) prefetch_begin(cur, knext);
                    if (s == 0) prefetch_c_part(p2, NP);
                    if (p2 == NP - 1) {
                        if (s < KSTEPS - NDB) prefetch_t(cur, s, tcur[s]);   // k-step s is done with tcur[s]
                        if (s == 0) {
#pragma unroll
                            for (int s2 = KSTEPS - NDB; s2 < KSTEPS; ++s2) prefetch_t(cur, s2, tnext[s2]);
                        }
                    }
                    if (s == KSTEPS - 1) stage_store_part(lds + (par ^ 1) * BUF, p2, NP);
                }
            }
            flip();
#pragma unroll
            for (int s = KSTEPS - NDB; s < KSTEPS; ++s) {
                tcur[s] = tnext[s];
                asm volatile("" : "+v"(tcur[s]));
            }
        }
        // ---- last chunk of the strip: tile outer, k-step inner, stores between the MFMAs -------
        {
            constexpr int knext = 0;
            K1P_MARK(99);
            const double* buf = lds + par * BUF + lq * LDJ + lr;
            double cf[2][KSTEPS];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) cf[0][s] = buf[s * 4 * LDJ];
            constexpr int NPT = NT >= 8 ? NT / 2 : (NT + 1) / 2, NST = NT >= 6 ? 3 : 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    acc[0][t] = mfma_f64(cf[t & 1][s], tcur[s].x, acc[0][t]);
                    acc[1][t] = mfma_f64(cf[t & 1][s], tcur[s].y, acc[1][t]);
                    if (t + 1 < NT) cf[(t + 1) & 1][s] = buf[s * 4 * LDJ + (t + 1) * 16];
                }
                if (t < NPT) prefetch_part(nxt, knext, t, NPT);
                if (t >= 1) store_tile(cur, t - 1);
                if (t >= NT - NST) stage_store_part(lds + (par ^ 1) * BUF, t - (NT - NST), NST);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_tile(cur, NT - 1);
            rotate();
            K1P_MARK(100);
        }
        ++k1_items;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[0][t] = acc[1][t] = d4{0.0, 0.0, 0.0, 0.0};
        cur = nxt;
        group += grid_x;
    }
    K1P_CLK_END
}

template <int NT, int KS>
int launch_pair_nt(const double* T, const double* Cm, double* out, long A, int K, int J, long B, int ldc,
                   int ngroups, int batch, long t_bs, long c_bs, long o_bs, hipStream_t st)
{
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int KC = 4 * KS;
    constexpr int CREG = (KC * LDJ + PTHREADS - 1) / PTHREADS;
    const size_t lds_bytes = (size_t)2 * CREG * PTHREADS * sizeof(double);
    const long nbt = (B + 31) / 32;
    const long n_items = A * nbt;
    const long ngroups_items = (n_items + PWAVES - 1) / PWAVES;
    long per_slice = 256 / ((long)ngroups * batch);   // persistent: one workgroup per CU
    if (per_slice < 1) per_slice = 1;
    const long nblocks = ngroups_items < per_slice ? ngroups_items : per_slice;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)contract_pair_kernel<NT, KS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            oovqe_set_error("mode_contract: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((contract_pair_kernel<NT, KS>),
                       dim3((unsigned)nblocks, (unsigned)ngroups, (unsigned)batch), dim3(PTHREADS), lds_bytes, st,
                       T, Cm, out, A, K, J, B, ldc, n_items, (int)nbt, t_bs, c_bs, o_bs);
    OOVQE_CHECK_LAUNCH("mode_contract (pair)");
    return 0;
}

template <int KS>
int launch_pair_group(int nt, const double* T, const double* Cm, double* out, long A, int K, int J, long B,
                      int ldc, int ngroups, int batch, long t_bs, long c_bs, long o_bs, hipStream_t st)
{
    switch (nt) {
#define OOVQE_CASE(n) \
    case n:           \
        return launch_pair_nt<n, KS>(T, Cm, out, A, K, J, B, ldc, ngroups, batch, t_bs, c_bs, o_bs, st);
        OOVQE_CASE(5) OOVQE_CASE(6) OOVQE_CASE(7) OOVQE_CASE(8) OOVQE_CASE(9) OOVQE_CASE(10)
        OOVQE_CASE(11) OOVQE_CASE(12) OOVQE_CASE(13)
#undef OOVQE_CASE
    }
    oovqe_set_error("mode_contract (pair): bad tile count %d", nt);
    return OOVQE_ERR_ARG;
}

}  // namespace

// Can this INNER contraction run on the two-strip kernel?  (Even row length and 16-byte aligned
// rows for the pair loads / stores, at least 5 tiles per wave, enough strips to fill the CUs.)
int oovqe_contract_pair_ok(const double* T, const double* out, long A, long B, int nt, int ngroups, int batch,
                           long t_bs, long o_bs)
{
    if ((B & 1) || ((size_t)T & 15) || ((size_t)out & 15)) return 0;
    if (batch > 1 && ((t_bs & 1) || (o_bs & 1))) return 0;
    if (nt < 5 || nt > 13) return 0;
    // 32-wide strips must not pad the row much more than 16-wide ones do (B = 200: 224 against 208
    // columns of MFMA work, measured 56.6 against 59.7 TFLOP/s)
    if (((B + 31) / 32) * 32 * 100 > ((B + 15) / 16) * 16 * 102) return 0;
    const long n_items = A * ((B + 31) / 32);
    return n_items * ngroups * batch >= 2 * 256 * PWAVES;
}

int oovqe_contract_pair_launch(const double* T, const double* Cm, double* out, long A, int K, int J, long B,
                               int ldc, int nt, int ngroups, int deep, int batch, long t_bs, long c_bs,
                               long o_bs, hipStream_t st)
{
    OOVQE_REQUIRE((double)A * (double)((B + 31) / 32) < 2.0e9, "mode_contract (pair): too many strips");
    return deep ? launch_pair_group<5>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, batch, t_bs, c_bs, o_bs, st)
                : launch_pair_group<3>(nt, T, Cm, out, A, K, J, B, ldc, ngroups, batch, t_bs, c_bs, o_bs, st);
}
