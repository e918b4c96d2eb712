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
//     needs 1 global load + 13 LDS reads: the kernel is MFMA-bound by construction;
//   * what keeps the matrix pipe fed in practice (round 2): the fragment read
//     of tile t for the next k-step right behind the MFMA of tile t; one descriptor per chunk and
//     the k-step as scalar offset; prefetch loads and LDS staging stores in parts between pairs of
//     MFMAs; the last chunk of a strip tile-outer with the tile's four stores (range-checked
//     buffer stores) behind the next tile's MFMAs; every wait for a prefetched load pinned inside
//     the straight-line code that issued it, where it is counted exactly.
//   INNER contractions with even B and many strips run on contract_pair.hip (two strips per wave).
#include "common.h"

// cache policy of the result stores / the T loads (buffer aux bits; 2 = nt: streaming)
#ifndef K1_AUX_ST
#define K1_AUX_ST 0
#endif
#ifndef K1_AUX_LD
#define K1_AUX_LD 0
#endif
#include "circuit_small.h"

// tools/k1_standalone.hip ablations (bit mask): 1 no T loads, 2 no stores, 4 no Cm staging,
// 8 no workgroup barrier, 16 every strip reads the first strip of T (cache hits), 32 every strip
// writes the first strip of out, 64 clock marks (with 512: around the loads and one tile's stores),
// 128 / 256 sixty-four dependent VALU / SALU instructions per k-step.  Never set in the library build.
#ifndef OOVQE_K1_PROBE
#define OOVQE_K1_PROBE 0
#endif

#if OOVQE_K1_PROBE & 64
// clock marks of wave 0 of workgroup 0 from its 4th strip on (chunk tops, epilogue begin / end)
__device__ long long g_k1_marks[128];
#define K1_MARK(code)                                                                          \
    do {                                                                                       \
        if (k1_items >= 3 && k1_m < 64 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
            g_k1_marks[2 * k1_m] = (code);                                                     \
            g_k1_marks[2 * k1_m + 1] = clock64();                                              \
        }                                                                                      \
        if (k1_items >= 3) ++k1_m;                                                             \
    } while (0)
#else
#define K1_MARK(code) do { } while (0)
#endif

namespace {

// K rows per LDS chunk = 4 * KS (KS MFMA k-steps): KS = 5 for long contractions, 3 for short ones
// (K is padded with zero rows of Cm to a multiple of the chunk, so a smaller chunk wastes fewer
// MFMAs when K is small, e.g. K = 43 -> 48 instead of 60).
// Eight waves per workgroup, one workgroup per CU (194 VGPRs: two waves per SIMD).  Four-wave
// workgroups, two per CU, so that the two waves of a SIMD do not share a barrier, measured slower
// (55.9 against 59.5 TFLOP/s at N = 200: the staging work per thread doubles).
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int NT, bool LAST, int KSTEPS, bool WIDE>
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
    constexpr int NW = NWAVES, NTH = NTHREADS;
    constexpr int KC = 4 * KSTEPS;
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int CHUNK = KC * LDJ;
    constexpr int CREG = (CHUNK + NTH - 1) / NTH;
    constexpr int BUF = CREG * NTH;      // chunk buffer, padded so staging stores need no guard
    // lds: [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int nchunks = (K + KC - 1) / KC;
    const long n_groups = (n_items + NW - 1) / NW;

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
        long te;        // element index where the strip's readable part of T ends: its slab (INNER: the
                        // zero-padded k-steps past K then read nothing instead of the next slab, whose
                        // values, times the zero rows of Cm, could turn an Inf into a NaN here) / T (LAST)
        long a, bcol;
        unsigned tvo;   // this lane's byte offset inside a k-step block, or OOB
        long ob, oe;    // element indices in out: the strip's first result, the end of its slab
        unsigned ovo;   // this lane's byte offset inside a result tile, or OOB
        bool active;
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto decode = [&](long group) -> Strip {
        Strip st;
        const long item = group * NW + wave_u;
        st.active = group < n_groups && item < n_items;
        st.a = 0;
        st.bcol = 0;
        st.tb = 0;
        bool tvalid = false;
        if (LAST) {
            st.a = item * 16;
            tvalid = st.active && st.a + lr < A;
            st.tb = st.a * (long)K;
            st.te = t_elems;
            st.tvo = tvalid ? (unsigned)((lr * (long)K + lq) * sizeof(double)) : OOB;
            st.ob = st.a * (long)J + j0;
            st.oe = A * (long)J;
            st.ovo = st.active ? (unsigned)((lq * (long)J + lr) * sizeof(double)) : OOB;
        } else {
            // 32-bit division (n_items < 2^31 is checked on the host)
            const unsigned ai = (unsigned)item / (unsigned)nbt;
            st.a = ai;
            const long bt = (long)((unsigned)item - ai * (unsigned)nbt);
            st.bcol = bt * 16;
            tvalid = st.active && st.bcol + lr < B;
            st.tb = st.a * (long)K * B + bt * 16;
            st.te = (st.a + 1) * (long)K * B;
            st.tvo = tvalid ? (unsigned)((lq * B + lr) * sizeof(double)) : OOB;
            st.ob = (st.a * (long)J + j0) * B + bt * 16;
            st.oe = (st.a * (long)J + J) * B;
            st.ovo = st.tvo;
        }
        if (!st.active) st.tb = 0, st.te = 0, st.ob = 0, st.oe = 0;
#if OOVQE_K1_PROBE & 16
        st.tb = 0;
#endif
#if OOVQE_K1_PROBE & 32
        st.oe -= st.ob;
        st.ob = 0;
#endif
        return st;
    };
    const long tstride = LAST ? 1 : B;

    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};

    double creg[CREG];
    double tcur[KSTEPS], tnext[KSTEPS];
#if OOVQE_K1_PROBE & 1
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) tnext[s] = 1.0;
#endif

    // Staging geometry of Cm, once per workgroup: byte offset of this thread's i-th element inside
    // a chunk (row kk, column j0 + jj), OOB for padding columns / rows; the chunk's first row is
    // added as a scalar offset and rows k >= K fall outside the descriptor (-> 0, the zero padding
    // the MFMAs rely on).
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
    auto stage_load = [&](int kbase) {
        const unsigned so = (unsigned)((long)kbase * ldc * sizeof(double));
#pragma unroll
        for (int i = 0; i < CREG; ++i)
            creg[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc_c, cvo[i], so, 0));
    };
    auto stage_store = [&](double* buf) {
#pragma unroll
        for (int i = 0; i < CREG; ++i) buf[tid + i * NTH] = creg[i];
    };
    // k-step s of a chunk: lane (lq, lr) reads T[.., kbase + 4s + lq, ..].  One descriptor per chunk
    // (T can be larger than the 4 GB a descriptor spans; it ends where T ends), the k-step inside
    // the chunk is the scalar offset of the load, clamped to the descriptor's range so that a step
    // past the end of T has nothing in range (the range check compares the lane offset with
    // num_records - soffset).  Strides too long for a 32-bit step offset (WIDE) rebase the
    // descriptor per k-step instead.
    const unsigned step_bytes = WIDE ? 0u : (unsigned)(4 * tstride * (long)sizeof(double));
    auto load_t = [&](const Strip& st, int kbase, double* dst) {
        const long e0 = st.tb + (long)kbase * tstride;   // wave-uniform
        long rem = (st.te - e0) * (long)sizeof(double);
        rem = clamp_u32(rem);
        if constexpr (!WIDE) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double*>(T) + e0, 0, (int)(unsigned)rem, 0x00020000);
#pragma unroll
            for (int s2 = 0; s2 < KSTEPS; ++s2) {
                unsigned so = (unsigned)s2 * step_bytes;
                so = so < (unsigned)rem ? so : (unsigned)rem;
                const unsigned tvo = (!LAST || kbase + 4 * s2 + lq < K) ? st.tvo : OOB;
                dst[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, tvo, so, K1_AUX_LD));
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < KSTEPS; ++s2) {
                const long e1 = e0 + (long)(4 * s2) * tstride;
                long rem1 = (st.te - e1) * (long)sizeof(double);
                rem1 = clamp_u32(rem1);
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double*>(T) + e1, 0, (int)(unsigned)rem1, 0x00020000);
                dst[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, st.tvo, 0, K1_AUX_LD));
            }
        }
    };

    // ---- prologue (once per workgroup) -------------------------------------------------------
    // One step of the workgroup = one staged K-chunk; a strip takes nchunks steps, the last of them
    // through the "last chunk" body below.  (Letting each wave begin its strips at its own phase of
    // the chunk cycle, so that the result stores of the eight waves fall on different steps, was
    // measured slower: 56.4 against 59.5 TFLOP/s at N = 200 -- a store costs the SIMD its ~70 issue
    // cycles wherever it is placed.)
    const long n_rounds = (n_groups - (long)blockIdx.x + grid_x - 1) / grid_x;   // strips per wave
    const long n_steps = n_rounds * nchunks;
    long group = blockIdx.x;       // strip group this wave begins next
    Strip cur = decode(group);
    group += grid_x;
    int cw = 0;                    // chunks summed into the current strip
    stage_load(0);
    load_t(cur, 0, tcur);
    stage_store(lds);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(tcur[s]));   // (wait here, see rotate)
    int par = 0;   // LDS buffer holding the chunk being consumed

    // The prefetch of the next chunk (Cm rows into registers, the strip's T values) and the LDS
    // staging stores, cut into np parts that the chunk bodies place between their groups of MFMAs.
    // Why parts: the two waves of a SIMD run the same instruction sequence in step, so a stretch
    // without MFMAs longer than the ~128 cycles the other wave's queued MFMA covers leaves the
    // matrix pipe idle (measured: 64 dependent VALU or SALU instructions per k-step cost their full
    // ~600 cycles).  14 loads at ~16 cycles of issue each plus their descriptor arithmetic in one
    // block were such a stretch.  The asm keeps the SALU arithmetic behind the MFMAs it follows in
    // the source; left alone it is hoisted to the top of the chunk, right after the barrier.
#if OOVQE_K1_PROBE & 64
    int k1_items = 0, k1_m = 0;
#endif
    constexpr int NPF = CREG + KSTEPS;   // prefetch items: CREG loads of Cm, KSTEPS loads of T
    int pf_kn = 0;
    unsigned pf_cso = 0, pf_rem = 0;
    __amdgpu_buffer_rsrc_t pf_tr = rsrc_c;
    auto prefetch_part = [&](const Strip& stn, int knext, int part, int np) {
        const int lo = part * NPF / np, hi = (part + 1) * NPF / np;
        if (part == 0) {
            pf_kn = knext;
            asm volatile("" : "+s"(pf_kn));
            pf_cso = (unsigned)((long)pf_kn * ldc * sizeof(double));
        }
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            if (q < lo || q >= hi) continue;
#if OOVQE_K1_PROBE & 512
            if (q == 0) K1_MARK(50);
            if (q == CREG) K1_MARK(51);
#endif
            if (q < CREG) {
#if !(OOVQE_K1_PROBE & 4)
                creg[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc_c, cvo[q], pf_cso, 0));
#endif
            } else {
#if !(OOVQE_K1_PROBE & 1)
                const int s2 = q - CREG;
                if (s2 == 0 || WIDE) {
                    const long e1 = stn.tb + (long)(pf_kn + (WIDE ? 4 * s2 : 0)) * tstride;   // wave-uniform
                    long rem = (stn.te - e1) * (long)sizeof(double);
                    rem = clamp_u32(rem);
                    pf_rem = (unsigned)rem;
                    pf_tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T) + e1, 0, (int)pf_rem, 0x00020000);
                }
                unsigned so = WIDE ? 0u : (unsigned)s2 * step_bytes;
                so = so < pf_rem ? so : pf_rem;
                // LAST: a row of T is followed by the next row, so the k-steps past K are masked per lane
                const unsigned tvo = (!LAST || pf_kn + 4 * s2 + lq < K) ? stn.tvo : OOB;
                tnext[s2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(pf_tr, tvo, so, K1_AUX_LD));
#if OOVQE_K1_PROBE & 512
                if (s2 == KSTEPS - 1) K1_MARK(52);
#endif
#endif
            }
        }
    };
    auto stage_store_part = [&](double* buf, int part, int np) {
#if !(OOVQE_K1_PROBE & 4)
        const int lo = part * CREG / np, hi = (part + 1) * CREG / np;
#pragma unroll
        for (int i = 0; i < CREG; ++i)
            if (i >= lo && i < hi) buf[tid + i * NTH] = creg[i];
#endif
    };
    // End of a chunk: the staged rows are visible, the prefetched T values become current.  The
    // asm makes the copy (and with it the wait for those loads) happen HERE: left alone, the copies
    // sink to the top of the next chunk, behind the loop header, where the wait can no longer be
    // counted exactly and would also wait for every store issued since.
    auto rotate = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#if !(OOVQE_K1_PROBE & 8)
        __syncthreads();
#endif
        par ^= 1;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            tcur[s] = tnext[s];
            asm volatile("" : "+v"(tcur[s]));
        }
    };
    // The results of tile t: four stores through a descriptor that ends where the strip's slab of
    // out ends (rows j >= J / a >= A fall outside and are dropped, as are the lanes of columns past
    // the edge: their offset is OOB); row 4 i of the tile is the scalar offset.
    const unsigned ostep_bytes = (unsigned)(4 * (LAST ? (long)J : B) * (long)sizeof(double));
    auto store_tile = [&](const Strip& st, int t) {
#if OOVQE_K1_PROBE & 2
        if (acc[0][0] != 1234.5678) return;
#endif
        const long e0 = st.ob + (long)t * (LAST ? 16 : 16 * B);   // wave-uniform
        long rem = (st.oe - e0) * (long)sizeof(double);
        rem = clamp_u32(rem);
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(out + e0, 0, (int)(unsigned)rem, 0x00020000);
        unsigned vo = st.ovo;
        if (LAST) vo = (j0 + t * 16 + lr < J) ? vo : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double v = acc[t][i];   // (bit_cast of the vector-element expression itself reads element 0)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, vo, (unsigned)i * ostep_bytes, K1_AUX_ST);
        }
    };

    int kg = 0;   // K-chunk of the current step
    for (long g = 0; g < n_steps; ++g) {
        kg = kg + 1 == nchunks ? 0 : kg + 1;
        const int knext = kg * KC;   // rows of the chunk to prefetch
        // ---- all chunks but the last: k-step outer, tile inner ----------------------------------
        // (rows k >= K of the staged Cm are zero, so the padded k-steps of a chunk add nothing.)
        // The Cm fragment of tile t for k-step s+1 is read from LDS right after the MFMA of tile t
        // for k-step s: a wave never issues a burst of LDS reads (eight waves doing so at the same
        // time fill the LDS queue and hold back the MFMAs behind them).  The global prefetch sits
        // behind the MFMAs of k-step 0 and the LDS staging stores among those of the last k-step.
        if (cw + 1 < nchunks) {
            K1_MARK(cw);
            ++cw;
            const double* buf = lds + par * BUF + lq * LDJ + lr;
            double cv[2][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) cv[0][t] = buf[t * 16];
            constexpr int NP = (NT + 1) / 2;   // groups of two tiles
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 2 * p2; t < 2 * p2 + 2 && t < NT; ++t)
                        acc[t] = LAST ? mfma_f64(tcur[s], cv[s & 1][t], acc[t])
                                      : mfma_f64(cv[s & 1][t], tcur[s], acc[t]);
#pragma unroll
                    for (int t = 2 * p2; t < 2 * p2 + 2 && t < NT; ++t)
                        if (s + 1 < KSTEPS) cv[(s + 1) & 1][t] = buf[(s + 1) * 4 * LDJ + t * 16];
                    if (s == 0) prefetch_part(cur, knext, p2, NP);
                    if (s == KSTEPS - 1) stage_store_part(lds + (par ^ 1) * BUF, p2, NP);
                }
#if OOVQE_K1_PROBE & 128
                {
                    int dummy = s;
#pragma unroll
                    for (int q = 0; q < 64; ++q) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
                    if (dummy == 12345678) tnext[0] += 1.0;
                }
#endif
#if OOVQE_K1_PROBE & 256
                {
                    int dummy = s;
#pragma unroll
                    for (int q = 0; q < 64; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy));
                    if (dummy == 12345678) tnext[0] += 1.0;
                }
#endif
            }
            rotate();
            continue;
        }
        // ---- last chunk of the strip: tile outer, k-step inner ---------------------------------
        // Tile t is complete after its KSTEPS MFMAs and its four stores are issued behind the MFMAs
        // of tile t+1: the 4 NT stores of a strip (eight waves reach this point together, and the
        // CU takes one 64-lane store every ~18 cycles) are spread over the whole chunk instead of
        // stalling the matrix pipe in a burst after it, and this straight-line stretch lets the
        // waits for the prefetched loads be counted exactly (loads older than the stores).
        const Strip nxt = decode(group);
        group += grid_x;
        cw = 0;
        {
            K1_MARK(99);
            const double* buf = lds + par * BUF + lq * LDJ + lr;
            double cf[2][KSTEPS];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) cf[0][s] = buf[s * 4 * LDJ];
            // prefetch parts on the first tiles, staging stores on the last ones (the Cm loads have
            // the tiles in between to arrive)
            constexpr int NPT = NT >= 8 ? NT / 2 : (NT + 1) / 2, NST = NT >= 6 ? 3 : 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    acc[t] = LAST ? mfma_f64(tcur[s], cf[t & 1][s], acc[t])
                                  : mfma_f64(cf[t & 1][s], tcur[s], acc[t]);
                    if (t + 1 < NT) cf[(t + 1) & 1][s] = buf[s * 4 * LDJ + (t + 1) * 16];
                }
                if (t < NPT) prefetch_part(nxt, knext, t, NPT);
                if constexpr (!WIDE) {
                    if (t >= 1) store_tile(cur, t - 1);
                }
                if (t >= NT - NST) stage_store_part(lds + (par ^ 1) * BUF, t - (NT - NST), NST);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!WIDE) store_tile(cur, NT - 1);
            rotate();
            K1_MARK(100);
        }
        if constexpr (WIDE) {
            // strides too long for 32-bit row offsets: plain stores with 64-bit addresses
            // (one running pointer + a scheduling fence per tile: otherwise all 4 NT 64-bit store
            // addresses are materialised at once on top of the accumulators)
            const long bcol = cur.bcol + lr;
            if (cur.active && bcol < B) {
                double* op = out + (cur.a * (long)J + j0 + lq) * B + bcol;
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
        K1_MARK(101);
#if OOVQE_K1_PROBE & 64
        ++k1_items;
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        cur = nxt;
    }
}

template <int NT, bool LAST, int KSTEPS, bool WIDE>
__global__ __launch_bounds__(NTHREADS, 2)
void contract_kernel(const double* __restrict__ T, const double* __restrict__ Cm,
                     double* __restrict__ out, long A, int K, int J, long B, int ldc,
                     long n_items, int nbt, long t_bs, long c_bs, long o_bs)
{
    extern __shared__ double lds[];
    contract_body<NT, LAST, KSTEPS, WIDE>(T, Cm, out, A, K, J, B, ldc, n_items, nbt, t_bs, c_bs,
                                                o_bs, lds, gridDim.x);
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
    contract_body<NT, LAST, KSTEPS, false>(T, Cm, out, A, K, J, B, ldc, n_items, nbt, t_bs, c_bs,
                                                 o_bs, lds, gridDim.x - 1);
}

// k-steps of a chunk as 32-bit scalar offsets of one descriptor: the last step's offset plus the
// largest lane offset (3 rows + 15 columns) has to stay below 2^32
bool step_offsets_fit(int ks, long B)
{
    const int rows = 4 * ks - 1 > 16 ? 4 * ks - 1 : 16;   // k-steps of a chunk; rows of a result tile
    return (double)rows * (double)B * 8.0 + 128.0 < 4294967296.0;
}

template <int NT, bool LAST, int KS, bool WIDE = false>
int launch_nt(const double* T, const double* Cm, double* out, long A, int K, int J, long B,
              int ldc, int ngroups, long n_items, int nbt, int batch, long t_bs, long c_bs, long o_bs,
              hipStream_t st, const oovqe_circuit_job_t* cj = nullptr)
{
    constexpr int LDJ = 16 * (NT | 1);
    constexpr int KC = 4 * KS;
    constexpr int NW = NWAVES, NTH = NTHREADS;
    constexpr int CREG = (KC * LDJ + NTH - 1) / NTH;
    size_t lds_bytes = (size_t)2 * CREG * NTH * sizeof(double);
    // persistent grid: at most ~2 workgroups per CU in total, each walks many item groups
    const long ngroups_items = (n_items + NW - 1) / NW;
    long per_slice = 512 / ((long)ngroups * batch);
    if (per_slice < 1) per_slice = 1;
    const long nblocks = ngroups_items < per_slice ? ngroups_items : per_slice;
    OOVQE_REQUIRE(WIDE || LAST || step_offsets_fit(KS, B), "mode_contract: stride too long for this kernel");
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
                               dim3(NTH), lds_bytes, st, T, Cm, out, A, K, J, B, ldc, n_items, nbt,
                               t_bs, c_bs, o_bs, *cj);
            OOVQE_CHECK_LAUNCH("mode_contract+circuit");
            return 0;
        }
    } else {
        OOVQE_REQUIRE(!cj, "mode_contract: this shape cannot host circuit workgroups");
    }
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)contract_kernel<NT, LAST, KS, WIDE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) {
            oovqe_set_error("mode_contract: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((contract_kernel<NT, LAST, KS, WIDE>),
                       dim3((unsigned)nblocks, (unsigned)ngroups, (unsigned)batch), dim3(NTH),
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

// Strides of T too long for scalar step offsets (B >= 48.8M elements): INNER, 12-row chunks, one
// descriptor per k-step
int launch_wide(int nt, const double* T, const double* Cm, double* out, long A, int K, int J, long B,
                int ldc, int ngroups, long n_items, int nbt, int batch, long t_bs, long c_bs, long o_bs,
                hipStream_t st)
{
    switch (nt) {
#define OOVQE_CASE(n) \
    case n:                                                                                    \
        return launch_nt<n, false, 3, true>(T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, \
                                            t_bs, c_bs, o_bs, st);
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
    if (oovqe_opt(OOVQE_OPT_K1_FORCE_NT) > 0 && oovqe_opt(OOVQE_OPT_K1_FORCE_NT) < nt) nt = oovqe_opt(OOVQE_OPT_K1_FORCE_NT);
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
    return K <= 48 && nt <= 4 && (last || step_offsets_fit(12, B));
}

int oovqe_mode_contract_batched_circ(const double* T, const double* Cm, double* out, long A, int K, int J,
                                     long B, int ldc, int last, int batch, long t_bs, long c_bs,
                                     long o_bs, hipStream_t st, const oovqe_circuit_job_t* cj);
int oovqe_contract_pair_ok(const double* T, const double* out, long A, long B, int nt, int ngroups, int batch,
                           long t_bs, long o_bs);
int oovqe_contract_pair_launch(const double* T, const double* Cm, double* out, long A, int K, int J, long B,
                               int ldc, int nt, int ngroups, int deep, int batch, long t_bs, long c_bs,
                               long o_bs, hipStream_t st);

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
    const bool deep = (pad20 <= pad12 || pad20 * 100 <= K * 105) && (last || step_offsets_fit(5, B));
    int rc;
    OOVQE_REQUIRE(!cj || (K <= 48 && nt <= 4 && (last || step_offsets_fit(12, B))),
                  "mode_contract: this shape cannot host circuit workgroups");
    if (!last && !cj && !(K <= 48 && nt <= 4) && step_offsets_fit(deep ? 5 : 3, B) && !oovqe_opt(OOVQE_OPT_K1_NO_PAIR) &&
        !oovqe_opt(OOVQE_OPT_K1_FORCE_WIDE) &&
        oovqe_contract_pair_ok(T, out, A, B, nt, ngroups, batch, t_bs, o_bs))
        // two 16-wide strips per wave (contract_pair.hip)
        rc = oovqe_contract_pair_launch(T, Cm, out, A, K, J, B, ldc, nt, ngroups, deep ? 1 : 0, batch, t_bs, c_bs,
                                        o_bs, st);
    else if (!last && (!step_offsets_fit(3, B) || oovqe_opt(OOVQE_OPT_K1_FORCE_WIDE)))
        rc = launch_wide(nt, T, Cm, out, A, K, J, B, ldc, ngroups, n_items, nbt, batch, t_bs, c_bs, o_bs, st);
    else if (K <= 48 && nt <= 4 && (last || step_offsets_fit(12, B)))
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

extern "C" int oovqe_matmul_nn_batch(const double* A, const double* B, int M, int K, int N, int batch,
                                     double* out, oovqe_stream_t stream)
{
    // out[b] = A[b] B[b] for a stack of independent products in ONE launch (the batch index is a grid
    // dimension of the LAST-mode contraction): mo_coeff[g] = S^-1/2[g] C_oao[g] of a stack of geometries
    OOVQE_REQUIRE(A && B && out, "matmul_nn_batch: null pointer");
    OOVQE_REQUIRE(M >= 1 && K >= 1 && N >= 1 && batch >= 1 && batch <= 65535, "matmul_nn_batch: bad sizes");
    return oovqe_mode_contract_batched(A, B, out, M, K, N, 1, N, 1, batch, (long)M * K, (long)K * N, (long)M * N,
                                       (hipStream_t)stream);
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
