// Statevector simulation of excitation-gate circuits (UCCD / kUpCCD / UCCSD) and RDMs.
//
// The reference hands a UCCD / kUpCCD operator to PennyLane's default.qubit, which decomposes
// every FermionicDoubleExcitation into ~150 one-/two-qubit gates (src/auto_oo/ansatze/uccd.py:
// 105-114, kUpCCD.py:118-130, pqc.py:69-76).  Here each excitation is ONE sparse Givens pass
// over the 2^n real amplitudes (closed form: include/oovqe.h, oovqe_gate_t); the state never
// becomes complex.  Tangent states d psi / d theta_k come from the same pass with the k-th gate
// replaced by its derivative (forward mode).
#include "common.h"
#include "circuit_small.h"

namespace {

constexpr int CIRC_THREADS = 256;
constexpr int LDS_STATE_MAX = 8192;   // amplitudes kept in LDS (64 KiB); larger states live in HBM/L2

// Apply one gate (or its theta-derivative) to `st` (LDS or global), D = 2^n amplitudes.
__device__ void apply_gate(double* st, uint32_t D, const oovqe_gate_t& g, double c, double s,
                           bool deriv)
{
    const uint32_t fm = g.mask_hi | g.mask_lo;
    const uint32_t npairs = D >> g.nfix;
    if (!deriv) {
        for (uint32_t t = threadIdx.x; t < npairs; t += CIRC_THREADS) {
            const uint32_t x = deposit(t, g) | g.mask_hi;
            const uint32_t y = x ^ fm;
            const double pi = (__popc(x & g.mask_par) & 1) ? -1.0 : 1.0;
            const double ax = st[x], ay = st[y];
            st[x] = c * ax + pi * s * ay;
            st[y] = c * ay - pi * s * ax;
        }
    } else {
        // d/dtheta [c, pi s; -pi s, c] = (sign/2) [-s, pi c; -pi c, -s]; identity part -> 0
        const double h = 0.5 * (double)g.sign;
        for (uint32_t t = threadIdx.x; t < npairs; t += CIRC_THREADS) {
            const uint32_t x = deposit(t, g) | g.mask_hi;
            const uint32_t y = x ^ fm;
            const double pi = (__popc(x & g.mask_par) & 1) ? -1.0 : 1.0;
            const double ax = st[x], ay = st[y];
            st[x] = h * (-s * ax + pi * c * ay);
            st[y] = h * (-s * ay - pi * c * ax);
        }
        for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) {
            const uint32_t f = x & fm;
            if (f != g.mask_hi && f != g.mask_lo) st[x] = 0.0;
        }
    }
}

// Run the whole circuit on working buffer w (LDS or global); dgate = index of the gate to
// differentiate, or -1.
__device__ void run_circuit(double* w, uint32_t D, uint32_t init_index, const double* th,
                            const oovqe_gate_t* __restrict__ gates, int n_gates, int dgate)
{
    for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) w[x] = (x == init_index) ? 1.0 : 0.0;
    __syncthreads();
    for (int g = 0; g < n_gates; ++g) {
        const oovqe_gate_t gt = gates[g];
        if (gt.theta_idx < 0) continue;   // fixed zero angle: identity (never differentiated)
        double s, c;
        sincos(0.5 * (double)gt.sign * th[gt.theta_idx], &s, &c);
        apply_gate(w, D, gt, c, s, g == dgate);
        __syncthreads();
    }
}

// grid = batch * (1 + n_tan); block b*(1+n_tan) + 0 -> psi, + (1+k) -> d psi / d theta_k.
// use_lds: the working state lives in LDS (D <= LDS_STATE_MAX) and is copied / accumulated to
// the output; otherwise the output vector itself is the working state (then a parameter may
// drive at most one gate - checked on the host).
__global__ __launch_bounds__(CIRC_THREADS)
void circuit_kernel(const double* __restrict__ theta, int n_theta,
                    const oovqe_gate_t* __restrict__ gates, int n_gates, int n_qubits,
                    uint32_t init_index, int n_tan, double* __restrict__ psi,
                    double* __restrict__ dpsi, int use_lds)
{
    extern __shared__ double lds[];
    const uint32_t D = 1u << n_qubits;
    const int per = 1 + n_tan;
    const int b = blockIdx.x / per;
    const int k = blockIdx.x % per - 1;
    const double* th = theta + (size_t)b * n_theta;
    double* dst = (k < 0) ? psi + (size_t)b * D : dpsi + ((size_t)b * n_theta + k) * D;

    if (k < 0) {
        if (use_lds) {
            run_circuit(lds, D, init_index, th, gates, n_gates, -1);
            for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) dst[x] = lds[x];
        } else {
            run_circuit(dst, D, init_index, th, gates, n_gates, -1);
        }
        return;
    }
    // tangent k = sum over the gates driven by theta_k of (circuit with that gate differentiated)
    bool first = true;
    for (int g = 0; g < n_gates; ++g) {
        if (gates[g].theta_idx != k) continue;
        if (use_lds) {
            run_circuit(lds, D, init_index, th, gates, n_gates, g);
            for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS)
                dst[x] = first ? lds[x] : dst[x] + lds[x];
            __syncthreads();
        } else if (first) {
            run_circuit(dst, D, init_index, th, gates, n_gates, g);
        }
        first = false;
    }
    if (first)
        for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) dst[x] = 0.0;
}

// ------------------------------------------------------------------------------------------
// RDMs.  V[pq][x] = (E_pq psi)[x],  E_pq = a+_{2p} a_{2q} + a+_{2p+1} a_{2q+1}
// (utils/active_space.py:46-52; Jordan-Wigner, spin orbital j = wire j = bit n-1-j, sign =
// parity of the occupied modes strictly between).  Then
//   gamma[p,q]     = bra . V_ket[pq]
//   Gamma[p,q,r,s] = V_bra[qp] . V_ket[rs] - delta_qr gamma[p,s]          (pqc.py:213-217)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void epq_apply_kernel(const double* __restrict__ bra, const double* __restrict__ ket, int n_qubits,
                      int ncas, double* __restrict__ V)
{
    // grid: (x-chunks, 2*ncas^2, batch); V layout [batch][2][ncas^2][D]  (0 = bra, 1 = ket)
    const uint32_t D = 1u << n_qubits;
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= D) return;
    const int na2 = ncas * ncas;
    const int which = blockIdx.y / na2;
    const int pq = blockIdx.y - which * na2;
    const int p = pq / ncas, q = pq - p * ncas;
    const int b = blockIdx.z;
    const double* src = (which == 0 ? bra : ket) + (size_t)b * D;
    double acc = 0.0;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int P = 2 * p + sp, Q = 2 * q + sp;
        const uint32_t bP = 1u << (n_qubits - 1 - P), bQ = 1u << (n_qubits - 1 - Q);
        if (p == q) {
            if (x & bP) acc += src[x];
        } else if ((x & bP) && !(x & bQ)) {
            // output x has P occupied, Q empty; source had Q occupied, P empty
            const uint32_t hi = bP > bQ ? bP : bQ, lo = bP > bQ ? bQ : bP;
            const uint32_t between = (hi - 1u) & ~((lo << 1) - 1u);
            const double sgn = (__popc(x & between) & 1) ? -1.0 : 1.0;
            acc += sgn * src[x ^ (bP | bQ)];
        }
    }
    V[(((size_t)b * 2 + which) * na2 + pq) * D + x] = acc;
}

__global__ __launch_bounds__(256)
void rdm_gram_kernel(const double* __restrict__ bra, const double* __restrict__ V, int n_qubits,
                     int ncas, double* __restrict__ gamma, double* __restrict__ Gamma)
{
    // grid: (ncas^2 [pq], batch).  One block produces gamma[pq] and the row Gamma[pq, :].
    __shared__ double scratch[256];
    __shared__ double gam_row[64];   // gamma[p, s] for s < ncas (ncas <= 64)
    const uint32_t D = 1u << n_qubits;
    const int na2 = ncas * ncas;
    const int pq = blockIdx.x, b = blockIdx.y;
    const int p = pq / ncas, q = pq - p * ncas;
    const int tid = threadIdx.x;
    const double* br = bra + (size_t)b * D;
    const double* Vb = V + ((size_t)b * 2 + 0) * na2 * D;
    const double* Vk = V + ((size_t)b * 2 + 1) * na2 * D;

    auto reduce = [&](double v) -> double {
        scratch[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        const double r = scratch[0];
        __syncthreads();
        return r;
    };

    // gamma[p, s] for all s (needed for the delta_qr term), gamma[p,q] written by this block
    for (int s = 0; s < ncas; ++s) {
        const double* vk = Vk + (size_t)(p * ncas + s) * D;
        double part = 0.0;
        for (uint32_t x = tid; x < D; x += 256) part += br[x] * vk[x];
        const double g = reduce(part);
        if (tid == 0) gam_row[s] = g;
    }
    __syncthreads();
    if (tid == 0) gamma[(size_t)b * na2 + pq] = gam_row[q];

    const double* vb = Vb + (size_t)(q * ncas + p) * D;   // E_qp bra
    for (int rs = 0; rs < na2; ++rs) {
        const double* vk = Vk + (size_t)rs * D;
        double part = 0.0;
        for (uint32_t x = tid; x < D; x += 256) part += vb[x] * vk[x];
        const double g = reduce(part);
        if (tid == 0) {
            const int r = rs / ncas, s = rs - r * ncas;
            Gamma[((size_t)b * na2 + pq) * na2 + rs] = g - (q == r ? gam_row[s] : 0.0);
        }
    }
}

// Small registers (D (2 + 2 a^2) doubles fit LDS): E_pq applied to bra and ket and all a^2 + a^4 scalar
// products by ONE workgroup per (bra, ket) pair, everything in LDS.  (The two kernels above start
// 2 a^2 + a^2 workgroups per pair and reduce every scalar product over a whole workgroup: at 6 qubits
// that is 12 tree reductions of 64 numbers per workgroup -- 150 us for the 2560 pairs of a 64-geometry
// circuit Hessian against ~10 us here.)
__global__ __launch_bounds__(256)
void rdms_small_kernel(const double* __restrict__ bra, const double* __restrict__ ket, int n_qubits,
                       int ncas, double* __restrict__ gamma, double* __restrict__ Gamma)
{
    extern __shared__ double lds[];
    const uint32_t D = 1u << n_qubits;
    const int LDV = (int)D + 1;              // odd pitch: rows on different banks
    const int na2 = ncas * ncas;
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x;
    double* vec = lds;                       // [2][LDV]        bra, ket
    double* V = vec + 2 * LDV;               // [2][na2][LDV]   E_pq bra, E_pq ket
    double* gam = V + (size_t)2 * na2 * LDV; // [na2]
    for (uint32_t x = tid; x < D; x += 256) {
        vec[x] = bra[b * D + x];
        vec[LDV + x] = ket[b * D + x];
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < 2u * na2 * D; idx += 256) {
        const uint32_t x = idx & (D - 1);
        const int row = (int)(idx >> n_qubits);
        const int which = row / na2, pq = row - which * na2;
        const int p = pq / ncas, q = pq - p * ncas;
        const double* src = vec + which * LDV;
        double acc = 0.0;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int P = 2 * p + sp, Q = 2 * q + sp;
            const uint32_t bP = 1u << (n_qubits - 1 - P), bQ = 1u << (n_qubits - 1 - Q);
            if (p == q) {
                if (x & bP) acc += src[x];
            } else if ((x & bP) && !(x & bQ)) {
                const uint32_t hi = bP > bQ ? bP : bQ, lo = bP > bQ ? bQ : bP;
                const uint32_t between = (hi - 1u) & ~((lo << 1) - 1u);
                const double sgn = (__popc(x & between) & 1) ? -1.0 : 1.0;
                acc += sgn * src[x ^ (bP | bQ)];
            }
        }
        V[(size_t)row * LDV + x] = acc;
    }
    __syncthreads();
    const double* Vb = V;
    const double* Vk = V + (size_t)na2 * LDV;
    for (int pq = tid; pq < na2; pq += 256) {
        const double* vk = Vk + (size_t)pq * LDV;
        double a0 = 0.0, a1 = 0.0;
        for (uint32_t x = 0; x < D; x += 2) {
            a0 += vec[x] * vk[x];
            a1 += vec[x + 1] * vk[x + 1];
        }
        gam[pq] = a0 + a1;
        gamma[b * na2 + pq] = a0 + a1;
    }
    __syncthreads();
    for (int idx = tid; idx < na2 * na2; idx += 256) {
        const int pq = idx / na2, rs = idx - pq * na2;
        const int p = pq / ncas, q = pq - p * ncas;
        const int r = rs / ncas, s2 = rs - r * ncas;
        const double* vb = Vb + (size_t)(q * ncas + p) * LDV;     // E_qp bra
        const double* vk = Vk + (size_t)rs * LDV;
        double a0 = 0.0, a1 = 0.0;
        for (uint32_t x = 0; x < D; x += 2) {
            a0 += vb[x] * vk[x];
            a1 += vb[x + 1] * vk[x + 1];
        }
        Gamma[(b * na2 + pq) * na2 + rs] = (a0 + a1) - (q == r ? gam[p * ncas + s2] : 0.0);
    }
}

// Apply E_pq to nvec vectors per batch element: vec layout [batch][nvec][D] given as psi (vector 0)
// and dpsi (vectors 1..nvec-1);  V layout [batch][nvec][ncas^2][D].
__global__ __launch_bounds__(256)
void epq_apply_multi_kernel(const double* __restrict__ psi, const double* __restrict__ dpsi,
                            int n_qubits, int ncas, int nvec, double* __restrict__ V)
{
    const uint32_t D = 1u << n_qubits;
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= D) return;
    const int na2 = ncas * ncas;
    const int v = blockIdx.y / na2;
    const int pq = blockIdx.y - v * na2;
    const int p = pq / ncas, q = pq - p * ncas;
    const int b = blockIdx.z;
    const double* src = (v == 0) ? psi + (size_t)b * D
                                 : dpsi + ((size_t)b * (nvec - 1) + (v - 1)) * D;
    double acc = 0.0;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int P = 2 * p + sp, Q = 2 * q + sp;
        const uint32_t bP = 1u << (n_qubits - 1 - P), bQ = 1u << (n_qubits - 1 - Q);
        if (p == q) {
            if (x & bP) acc += src[x];
        } else if ((x & bP) && !(x & bQ)) {
            const uint32_t hi = bP > bQ ? bP : bQ, lo = bP > bQ ? bQ : bP;
            const uint32_t between = (hi - 1u) & ~((lo << 1) - 1u);
            const double sgn = (__popc(x & between) & 1) ? -1.0 : 1.0;
            acc += sgn * src[x ^ (bP | bQ)];
        }
    }
    V[(((size_t)b * nvec + v) * na2 + pq) * D + x] = acc;
}

// RDMs of psi (set 0) and their theta-derivatives (set k >= 1):
//   d gamma_k[pq]   = dpsi_k . V0[pq] + psi . Vk[pq]
//   d Gamma_k[pqrs] = Vk[qp] . V0[rs] + V0[qp] . Vk[rs] - delta_qr d gamma_k[ps]
// grid: (ncas^2 [pq], nvec [k], batch)
__global__ __launch_bounds__(256)
void rdm_tangent_gram_kernel(const double* __restrict__ psi, const double* __restrict__ dpsi,
                             const double* __restrict__ V, int n_qubits, int ncas, int nvec,
                             double* __restrict__ gamma, double* __restrict__ Gamma)
{
    __shared__ double scratch[256];
    __shared__ double gam_row[64];
    const uint32_t D = 1u << n_qubits;
    const int na2 = ncas * ncas;
    const int pq = blockIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int p = pq / ncas, q = pq - p * ncas;
    const int tid = threadIdx.x;
    const double* ps = psi + (size_t)b * D;
    const double* dk = (k == 0) ? ps : dpsi + ((size_t)b * (nvec - 1) + (k - 1)) * D;
    const double* V0 = V + ((size_t)b * nvec + 0) * na2 * D;
    const double* Vk = V + ((size_t)b * nvec + k) * na2 * D;

    auto reduce = [&](double v) -> double {
        scratch[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        const double r = scratch[0];
        __syncthreads();
        return r;
    };

    for (int s = 0; s < ncas; ++s) {
        const double* v0 = V0 + (size_t)(p * ncas + s) * D;
        const double* vk = Vk + (size_t)(p * ncas + s) * D;
        double part = 0.0;
        if (k == 0) {
            for (uint32_t x = tid; x < D; x += 256) part += ps[x] * v0[x];
        } else {
            for (uint32_t x = tid; x < D; x += 256) part += dk[x] * v0[x] + ps[x] * vk[x];
        }
        const double g = reduce(part);
        if (tid == 0) gam_row[s] = g;
    }
    __syncthreads();
    const size_t set = (size_t)b * nvec + k;
    if (tid == 0) gamma[set * na2 + pq] = gam_row[q];

    const double* b0 = V0 + (size_t)(q * ncas + p) * D;   // E_qp psi
    const double* bk = Vk + (size_t)(q * ncas + p) * D;   // E_qp dpsi_k
    for (int rs = 0; rs < na2; ++rs) {
        const double* k0 = V0 + (size_t)rs * D;
        const double* kk = Vk + (size_t)rs * D;
        double part = 0.0;
        if (k == 0) {
            for (uint32_t x = tid; x < D; x += 256) part += b0[x] * k0[x];
        } else {
            for (uint32_t x = tid; x < D; x += 256) part += bk[x] * k0[x] + b0[x] * kk[x];
        }
        const double g = reduce(part);
        if (tid == 0) {
            const int r = rs / ncas, s = rs - r * ncas;
            Gamma[(set * na2 + pq) * na2 + rs] = g - (q == r ? gam_row[s] : 0.0);
        }
    }
}

// Small active spaces: state + tangents + E_pq applications + all RDM sets in ONE workgroup
// (circuit_small.h).  The same body also runs as extra workgroups of a K1 launch (contract.hip).
__global__ __launch_bounds__(SMALL_THREADS)
void circuit_rdm_small_kernel(const double* __restrict__ theta, int n_theta,
                              const oovqe_gate_t* __restrict__ gates, int n_gates, int n_qubits,
                              int ncas, uint32_t init_index, int n_tan, double* __restrict__ psi_out,
                              double* __restrict__ dpsi_out, double* __restrict__ gamma,
                              double* __restrict__ Gamma, int batch, const double* __restrict__ h_ao,
                              const double* __restrict__ C, int N, double* __restrict__ Wpre)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= batch) {
        // Extra workgroups of this launch (they run beside the circuit workgroups: the launch is one short
        // resident round): W = C^T h_ao of geometry blockIdx.x - batch for the Fock stage's panel kernel, which
        // otherwise stages all of h_ao in EVERY panel workgroup to form its own rows.  The same two-accumulator
        // sums as there: the same bits.  h_ao and C through LDS (2 N^2 doubles, N <= 48).
        const int g = (int)blockIdx.x - batch;
        const double* __restrict__ hg = h_ao + (size_t)g * N * N;
        const double* __restrict__ Cg = C + (size_t)g * N * N;
        double* __restrict__ Wg = Wpre + (size_t)g * N * N;
        double* hs = lds;
        double* cs = lds + (size_t)N * N;
        constexpr int IT = (48 * 48 + SMALL_THREADS - 1) / SMALL_THREADS;
        double rh[IT], rc[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + it * SMALL_THREADS;
            const int ic = idx < N * N ? idx : 0;
            rh[it] = hg[ic];
            rc[it] = Cg[ic];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + it * SMALL_THREADS;
            if (idx < N * N) { hs[idx] = rh[it]; cs[idx] = rc[it]; }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * N; idx += SMALL_THREADS) {
            const int n = idx / N, q = idx - n * N;
            double a0 = 0.0, a1 = 0.0;
            int p = 0;
            for (; p + 1 < N; p += 2) {
                a0 += cs[p * N + n] * hs[p * N + q];
                a1 += cs[(p + 1) * N + n] * hs[(p + 1) * N + q];
            }
            if (p < N) a0 += cs[p * N + n] * hs[p * N + q];
            Wg[idx] = a0 + a1;
        }
        return;
    }
    circuit_rdm_small_body(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index, n_tan, psi_out,
                           dpsi_out, gamma, Gamma, (int)blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// Second tangents d^2 psi / d theta_j d theta_k (for the circuit-circuit Hessian, reference
// oo_pqc.py:103-111 = torch.autograd.functional.hessian through the simulator).
// mode 0: gate, 1: d/dtheta, 2: d^2/dtheta^2 = -(1/4) gate on the pair subspace, 0 elsewhere.
// ------------------------------------------------------------------------------------------
__device__ void apply_gate_mode(double* st, uint32_t D, const oovqe_gate_t& g, double c, double s,
                                int mode)
{
    const uint32_t fm = g.mask_hi | g.mask_lo;
    const uint32_t npairs = D >> g.nfix;
    const double h = 0.5 * (double)g.sign;
    for (uint32_t t = threadIdx.x; t < npairs; t += CIRC_THREADS) {
        const uint32_t x = deposit(t, g) | g.mask_hi;
        const uint32_t y = x ^ fm;
        const double pi = (__popc(x & g.mask_par) & 1) ? -1.0 : 1.0;
        const double ax = st[x], ay = st[y];
        const double nx = c * ax + pi * s * ay, ny = c * ay - pi * s * ax;
        if (mode == 0) {
            st[x] = nx;
            st[y] = ny;
        } else if (mode == 1) {
            st[x] = h * (-s * ax + pi * c * ay);
            st[y] = h * (-s * ay - pi * c * ax);
        } else {
            st[x] = -0.25 * nx;
            st[y] = -0.25 * ny;
        }
    }
    if (mode != 0) {
        for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) {
            const uint32_t f = x & fm;
            if (f != g.mask_hi && f != g.mask_lo) st[x] = 0.0;
        }
    }
}

// grid = npairs; pair (j,k): out[pair] = sum over gates g1 driven by theta_j, g2 driven by theta_k
// of the circuit with g1 and g2 differentiated (g1 == g2: second derivative of that gate).
__global__ __launch_bounds__(CIRC_THREADS)
void circuit_second_tangent_kernel(const double* __restrict__ theta, int n_theta,
                                   const oovqe_gate_t* __restrict__ gates, int n_gates, int n_qubits,
                                   uint32_t init_index, const int32_t* __restrict__ pairs,
                                   double* __restrict__ out, double* __restrict__ scratch)
{
    const uint32_t D = 1u << n_qubits;
    const int j = pairs[2 * blockIdx.x], k = pairs[2 * blockIdx.x + 1];
    // blockIdx.y = element of a batch: theta [batch][n_theta], out / scratch [batch][n_pairs][D]
    theta += (size_t)blockIdx.y * n_theta;
    double* dst = out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * D;
    double* w = scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * D;
    for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) dst[x] = 0.0;
    __syncthreads();
    for (int g1 = 0; g1 < n_gates; ++g1) {
        if (gates[g1].theta_idx != j) continue;
        for (int g2 = 0; g2 < n_gates; ++g2) {
            if (gates[g2].theta_idx != k) continue;
            for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) w[x] = (x == init_index) ? 1.0 : 0.0;
            __syncthreads();
            for (int g = 0; g < n_gates; ++g) {
                const oovqe_gate_t gt = gates[g];
                if (gt.theta_idx < 0) continue;
                double sn, cs;
                sincos(0.5 * (double)gt.sign * theta[gt.theta_idx], &sn, &cs);
                const int mode = (g == g1 && g == g2) ? 2 : ((g == g1 || g == g2) ? 1 : 0);
                apply_gate_mode(w, D, gt, cs, sn, mode);
                __syncthreads();
            }
            for (uint32_t x = threadIdx.x; x < D; x += CIRC_THREADS) dst[x] += w[x];
            __syncthreads();
        }
    }
}

// H[j,k] = H[k,j] = sum over the 4 transition-RDM sets of pair (j,k) of c1.gamma + c2.Gamma
__global__ __launch_bounds__(256)
void circuit_hessian_kernel(const double* __restrict__ gamma, const double* __restrict__ Gamma,
                            const double* __restrict__ c1, const double* __restrict__ c2, int ncas,
                            const int32_t* __restrict__ pairs, int n_theta, double* __restrict__ H,
                            long c1_bs, long c2_bs, long ldh, long h_bs)
{
    __shared__ double red[256];
    const int na2 = ncas * ncas, na4 = na2 * na2;
    const int pr = blockIdx.x, tid = threadIdx.x;
    // blockIdx.y = element of a batch: RDM sets [batch][n_pairs][4], c1 / c2 / H with batch strides
    const size_t prb = (size_t)blockIdx.y * gridDim.x + pr;
    c1 += (size_t)blockIdx.y * c1_bs;
    c2 += (size_t)blockIdx.y * c2_bs;
    H += (size_t)blockIdx.y * h_bs;
    double acc = 0.0;
    for (int set = 0; set < 4; ++set) {
        const double* g1 = gamma + (prb * 4 + set) * na2;
        const double* g2 = Gamma + (prb * 4 + set) * na4;
        for (int i = tid; i < na2; i += 256) acc += c1[i] * g1[i];
        for (int i = tid; i < na4; i += 256) acc += c2[i] * g2[i];
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        const int j = pairs[2 * pr], k = pairs[2 * pr + 1];
        H[(size_t)j * ldh + k] = red[0];
        H[(size_t)k * ldh + j] = red[0];
    }
}

// operand lists of the theta-theta Hessian: per pair (j,k) the four (bra, ket) combinations
// (psi_jk, psi), (psi_j, psi_k), (psi_k, psi_j), (psi, psi_jk); grid (ceil(D/256), 4 n_pairs)
__global__ __launch_bounds__(256)
void hessian_operands_kernel(const double* __restrict__ psi, const double* __restrict__ dpsi,
                             const double* __restrict__ psi2, const int32_t* __restrict__ pairs,
                             unsigned D, int n_theta, double* __restrict__ bra, double* __restrict__ ket)
{
    const unsigned x = blockIdx.x * 256 + threadIdx.x;
    if (x >= D) return;
    const int row = blockIdx.y, pr = row >> 2, set = row & 3;
    // blockIdx.z = element of a batch: psi [batch][D], dpsi [batch][n_theta][D], psi2 [batch][n_pairs][D]
    const size_t b = blockIdx.z, n_rows = gridDim.y;
    psi += b * D;
    dpsi += b * n_theta * D;
    psi2 += b * (n_rows >> 2) * D;
    const int j = pairs[2 * pr], k = pairs[2 * pr + 1];
    const double p0 = psi[x], p2 = psi2[(size_t)pr * D + x];
    const double pj = dpsi[(size_t)j * D + x], pk = dpsi[(size_t)k * D + x];
    bra[(b * n_rows + row) * D + x] = set == 0 ? p2 : set == 1 ? pj : set == 2 ? pk : p0;
    ket[(b * n_rows + row) * D + x] = set == 0 ? p0 : set == 1 ? pk : set == 2 ? pj : p2;
}

}  // namespace

extern "C" int oovqe_circuit_state(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                   int n_gates, int n_qubits, uint32_t init_index, int batch,
                                   double* psi, double* dpsi, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && psi, "circuit_state: null pointer");
    OOVQE_REQUIRE(n_qubits >= 2 && n_qubits <= 26, "circuit_state: n_qubits=%d", n_qubits);
    OOVQE_REQUIRE(n_theta >= 1 && n_gates >= 1 && batch >= 1, "circuit_state: bad sizes");
    const uint32_t D = 1u << n_qubits;
    OOVQE_REQUIRE(init_index < D, "circuit_state: init_index out of range");
    const int use_lds = D <= (uint32_t)LDS_STATE_MAX;
    const int n_tan = dpsi ? n_theta : 0;
    const size_t lds_bytes = use_lds ? (size_t)D * sizeof(double) : 0;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)circuit_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_STATE_MAX * (int)sizeof(double));
        if (e != hipSuccess) {
            oovqe_set_error("circuit_state: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return OOVQE_ERR_HIP;
        }
        attr_done = true;
    }
    const long nblocks = (long)batch * (1 + n_tan);
    OOVQE_REQUIRE(nblocks <= 0x7fffffffL, "circuit_state: grid too large");
    hipLaunchKernelGGL(circuit_kernel, dim3((unsigned)nblocks), dim3(CIRC_THREADS), lds_bytes,
                       (hipStream_t)stream, theta, n_theta, gates, n_gates, n_qubits, init_index,
                       n_tan, psi, dpsi, use_lds);
    OOVQE_CHECK_LAUNCH("circuit_state");
    return 0;
}

extern "C" int oovqe_rdms(const double* bra, const double* ket, int n_qubits, int ncas, int batch,
                          double* gamma, double* Gamma, double* work, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(bra && ket && gamma && Gamma && work, "rdms: null pointer");
    OOVQE_REQUIRE(n_qubits == 2 * ncas && ncas >= 1 && ncas <= 13 && batch >= 1,
                  "rdms: bad sizes n_qubits=%d ncas=%d", n_qubits, ncas);
    const uint32_t D = 1u << n_qubits;
    const int na2 = ncas * ncas;
    hipStream_t st = (hipStream_t)stream;
    const size_t small_bytes = ((size_t)(2 + 2 * na2) * (D + 1) + na2) * sizeof(double);
    if (n_qubits >= 1 && small_bytes <= 150 * 1024) {
        if (small_bytes > 64 * 1024) {
            int rc_lds = oovqe_ensure_dynamic_lds((const void*)rdms_small_kernel, small_bytes);
            if (rc_lds) return rc_lds;
        }
        hipLaunchKernelGGL(rdms_small_kernel, dim3((unsigned)batch), dim3(256), small_bytes, st, bra, ket,
                           n_qubits, ncas, gamma, Gamma);
        OOVQE_CHECK_LAUNCH("rdms/small");
        return 0;
    }
    OOVQE_REQUIRE(batch <= 65535, "rdms: batch too large");
    hipLaunchKernelGGL(epq_apply_kernel, dim3((D + 255) / 256, 2 * na2, batch), dim3(256), 0, st,
                       bra, ket, n_qubits, ncas, work);
    OOVQE_CHECK_LAUNCH("rdms/epq_apply");
    hipLaunchKernelGGL(rdm_gram_kernel, dim3(na2, batch), dim3(256), 0, st, bra, work, n_qubits,
                       ncas, gamma, Gamma);
    OOVQE_CHECK_LAUNCH("rdms/gram");
    return 0;
}

extern "C" int oovqe_rdms_tangent(const double* psi, const double* dpsi, int n_qubits, int ncas,
                                  int n_tan, int batch, double* gamma, double* Gamma, double* work,
                                  oovqe_stream_t stream)
{
    OOVQE_REQUIRE(psi && gamma && Gamma && work, "rdms_tangent: null pointer");
    OOVQE_REQUIRE(n_tan == 0 || dpsi, "rdms_tangent: dpsi required when n_tan > 0");
    OOVQE_REQUIRE(n_qubits == 2 * ncas && ncas >= 1 && ncas <= 13 && batch >= 1 && n_tan >= 0,
                  "rdms_tangent: bad sizes n_qubits=%d ncas=%d", n_qubits, ncas);
    const uint32_t D = 1u << n_qubits;
    const int na2 = ncas * ncas;
    const int nvec = 1 + n_tan;
    OOVQE_REQUIRE(batch <= 65535 && (long)nvec * na2 <= 65535, "rdms_tangent: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(epq_apply_multi_kernel, dim3((D + 255) / 256, nvec * na2, batch), dim3(256),
                       0, st, psi, dpsi, n_qubits, ncas, nvec, work);
    OOVQE_CHECK_LAUNCH("rdms_tangent/epq_apply");
    hipLaunchKernelGGL(rdm_tangent_gram_kernel, dim3(na2, nvec, batch), dim3(256), 0, st, psi, dpsi,
                       work, n_qubits, ncas, nvec, gamma, Gamma);
    OOVQE_CHECK_LAUNCH("rdms_tangent/gram");
    return 0;
}

static size_t small_lds_bytes(int n_qubits, int ncas, int nvec, int n_gates)
{
    return oovqe_small_circuit_lds_bytes(n_qubits, ncas, nvec, n_gates);
}

extern "C" int oovqe_circuit_rdms_is_small(int n_qubits, int ncas, int nvec, int n_gates)
{
    return n_qubits <= 10 && small_lds_bytes(n_qubits, ncas, nvec, n_gates) <= 150 * 1024;
}

// cas.hip: the same with W = C^T h_ao [batch][N][N] of every geometry formed by extra workgroups of the launch
// (small circuits only: the caller has checked oovqe_circuit_rdms_is_small; N <= 48)
int oovqe_circuit_rdms_w(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int n_qubits,
                         int ncas, uint32_t init_index, int want_tangents, int batch, double* psi, double* dpsi,
                         double* gamma, double* Gamma, double* work, const double* h_ao, const double* C, int N,
                         double* Wpre, oovqe_stream_t stream);

extern "C" int oovqe_circuit_rdms(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                  int n_gates, int n_qubits, int ncas, uint32_t init_index,
                                  int want_tangents, int batch, double* psi, double* dpsi,
                                  double* gamma, double* Gamma, double* work, oovqe_stream_t stream)
{
    return oovqe_circuit_rdms_w(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index, want_tangents, batch, psi,
                                dpsi, gamma, Gamma, work, nullptr, nullptr, 0, nullptr, stream);
}

int oovqe_circuit_rdms_w(const double* theta, int n_theta, const oovqe_gate_t* gates, int n_gates, int n_qubits,
                         int ncas, uint32_t init_index, int want_tangents, int batch, double* psi, double* dpsi,
                         double* gamma, double* Gamma, double* work, const double* h_ao, const double* C, int N,
                         double* Wpre, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && gamma && Gamma, "circuit_rdms: null pointer");
    OOVQE_REQUIRE(n_qubits == 2 * ncas && ncas >= 1 && ncas <= 13, "circuit_rdms: bad sizes");
    OOVQE_REQUIRE(n_theta >= 1 && n_gates >= 1 && batch >= 1, "circuit_rdms: bad sizes");
    const int n_tan = want_tangents ? n_theta : 0;
    const int nvec = 1 + n_tan;
    size_t lds_bytes = small_lds_bytes(n_qubits, ncas, nvec, n_gates);
    OOVQE_REQUIRE(!Wpre || (h_ao && C && N >= 1 && N <= 48 && oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates) &&
                            batch <= 32767),
                  "circuit_rdms: W = C^T h rides with the one-workgroup circuit kernel (N <= 48)");
    if (Wpre && (size_t)2 * N * N * sizeof(double) > lds_bytes) lds_bytes = (size_t)2 * N * N * sizeof(double);
    if (oovqe_circuit_rdms_is_small(n_qubits, ncas, nvec, n_gates)) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)circuit_rdm_small_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               150 * 1024);
            if (e != hipSuccess) {
                oovqe_set_error("circuit_rdms: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return OOVQE_ERR_HIP;
            }
            attr_done = true;
        }
        hipLaunchKernelGGL(circuit_rdm_small_kernel, dim3(Wpre ? 2 * batch : batch), dim3(SMALL_THREADS), lds_bytes,
                           (hipStream_t)stream, theta, n_theta, gates, n_gates, n_qubits, ncas,
                           init_index, n_tan, psi, dpsi, gamma, Gamma, batch, h_ao, C, N, Wpre);
        OOVQE_CHECK_LAUNCH("circuit_rdms/small");
        return 0;
    }
    // general path: separate launches, vectors in HBM/L2
    OOVQE_REQUIRE(psi && work && (!want_tangents || dpsi),
                  "circuit_rdms: psi/dpsi/work buffers required for this size");
    int rc = oovqe_circuit_state(theta, n_theta, gates, n_gates, n_qubits, init_index, batch, psi,
                                 want_tangents ? dpsi : nullptr, stream);
    if (rc) return rc;
    return oovqe_rdms_tangent(psi, want_tangents ? dpsi : nullptr, n_qubits, ncas, n_tan, batch,
                              gamma, Gamma, work, stream);
}

extern "C" int oovqe_circuit_second_tangents(const double* theta, int n_theta,
                                             const oovqe_gate_t* gates, int n_gates, int n_qubits,
                                             uint32_t init_index, const int32_t* pairs, int n_pairs,
                                             double* out, double* scratch, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && pairs && out && scratch, "second_tangents: null pointer");
    OOVQE_REQUIRE(n_qubits >= 2 && n_qubits <= 26 && n_pairs >= 1 && n_gates >= 1 && n_theta >= 1,
                  "second_tangents: bad sizes");
    hipLaunchKernelGGL(circuit_second_tangent_kernel, dim3(n_pairs), dim3(CIRC_THREADS), 0,
                       (hipStream_t)stream, theta, n_theta, gates, n_gates, n_qubits, init_index, pairs, out,
                       scratch);
    OOVQE_CHECK_LAUNCH("second_tangents");
    return 0;
}

extern "C" int oovqe_circuit_hessian_assemble(const double* gamma, const double* Gamma,
                                              const double* c1, const double* c2, int ncas,
                                              const int32_t* pairs, int n_pairs, int n_theta,
                                              double* H, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(gamma && Gamma && c1 && c2 && pairs && H, "circuit_hessian: null pointer");
    hipLaunchKernelGGL(circuit_hessian_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream,
                       gamma, Gamma, c1, c2, ncas, pairs, n_theta, H, 0L, 0L, (long)n_theta, 0L);
    OOVQE_CHECK_LAUNCH("circuit_hessian");
    return 0;
}


// d^2E/dtheta^2 for E = c0 + c1.gamma(theta) + c2.Gamma(theta) in ONE call (oo_pqc.py:103-111):
// state + tangents, second tangents, the 4 n_pairs transition-RDM operand pairs, transition RDMs,
// contraction -- five launches chained here instead of a dozen small tensor operations on the host
// side (0.26 -> ~0.07 ms for the 4 x 4 block of configs[1]).  pairs [n_pairs][2] (j <= k).
extern "C" int64_t oovqe_circuit_hessian_work_size(int n_theta, int n_qubits, int ncas, int n_pairs)
{
    const int64_t D = (int64_t)1 << n_qubits, na2 = (int64_t)ncas * ncas;
    // psi | dpsi | psi2 | scratch | bra | ket | gamma | Gamma | rdm work
    return D * (1 + n_theta + 2 * n_pairs + 8 * n_pairs) + 4 * n_pairs * (na2 + na2 * na2) +
           4 * n_pairs * 2 * na2 * D;
}

// batch elements stacked: theta [batch][n_theta]; c1 / c2 of element b at c1 + b * c1_bs, c2 + b * c2_bs;
// H element (j,k) of b at H[b * h_bs + j * ldh + k]; work: batch * oovqe_circuit_hessian_work_size()
int oovqe_circuit_hessian_batched_impl(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                       int n_gates, int n_qubits, int ncas, uint32_t init_index,
                                       const double* c1, const double* c2, long c1_bs, long c2_bs,
                                       const int32_t* pairs, int n_pairs, int batch, double* work, double* H,
                                       long ldh, long h_bs, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(theta && gates && c1 && c2 && pairs && work && H, "circuit_hessian: null pointer");
    OOVQE_REQUIRE(n_pairs >= 1 && 4 * n_pairs <= 65535, "circuit_hessian: n_pairs=%d", n_pairs);
    OOVQE_REQUIRE(batch >= 1, "circuit_hessian: batch=%d", batch);
    if ((long)batch * 4 * n_pairs > 65535 || batch > 65535) {
        // the launches below put (geometry, pair) into 16-bit grid dimensions: larger stacks go through in
        // chunks of geometries, one after the other on the stream, sharing the workspace
        const int chunk = 65535 / (4 * n_pairs);
        for (int b0 = 0; b0 < batch; b0 += chunk) {
            const int nbc = batch - b0 < chunk ? batch - b0 : chunk;
            int rcc = oovqe_circuit_hessian_batched_impl(theta + (size_t)b0 * n_theta, n_theta, gates, n_gates, n_qubits,
                                                         ncas, init_index, c1 + (long)b0 * c1_bs, c2 + (long)b0 * c2_bs,
                                                         c1_bs, c2_bs, pairs, n_pairs, nbc, work, H + (long)b0 * h_bs,
                                                         ldh, h_bs, stream);
            if (rcc) return rcc;
        }
        return 0;
    }
    const size_t D = (size_t)1 << n_qubits, na2 = (size_t)ncas * ncas, nb = (size_t)batch;
    double* psi = work;                                       // [G][D]
    double* dpsi = psi + nb * D;                              // [G][n_theta][D]
    double* psi2 = dpsi + nb * n_theta * D;                   // [G][n_pairs][D]
    double* scratch = psi2 + nb * n_pairs * D;                // [G][n_pairs][D]
    double* bra = scratch + nb * n_pairs * D;                 // [G][4 n_pairs][D]
    double* ket = bra + 4 * nb * n_pairs * D;
    double* g1 = ket + 4 * nb * n_pairs * D;                  // [G][4 n_pairs][a^2]
    double* g2 = g1 + 4 * nb * n_pairs * na2;                 // [G][4 n_pairs][a^4]
    double* rwork = g2 + 4 * nb * n_pairs * na2 * na2;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = oovqe_circuit_state(theta, n_theta, gates, n_gates, n_qubits, init_index, batch, psi, dpsi, stream)))
        return rc;
    OOVQE_REQUIRE(n_qubits >= 2 && n_qubits <= 26 && n_gates >= 1 && n_theta >= 1, "circuit_hessian: bad sizes");
    hipLaunchKernelGGL(circuit_second_tangent_kernel, dim3(n_pairs, batch), dim3(CIRC_THREADS), 0, st, theta,
                       n_theta, gates, n_gates, n_qubits, init_index, pairs, psi2, scratch);
    OOVQE_CHECK_LAUNCH("circuit_hessian/second_tangents");
    hipLaunchKernelGGL(hessian_operands_kernel,
                       dim3((unsigned)((D + 255) / 256), (unsigned)(4 * n_pairs), (unsigned)batch), dim3(256), 0,
                       st, psi, dpsi, psi2, pairs, (unsigned)D, n_theta, bra, ket);
    OOVQE_CHECK_LAUNCH("circuit_hessian/operands");
    if ((rc = oovqe_rdms(bra, ket, n_qubits, ncas, 4 * n_pairs * batch, g1, g2, rwork, stream))) return rc;
    hipLaunchKernelGGL(circuit_hessian_kernel, dim3(n_pairs, batch), dim3(256), 0, st, g1, g2, c1, c2, ncas,
                       pairs, n_theta, H, c1_bs, c2_bs, ldh, h_bs);
    OOVQE_CHECK_LAUNCH("circuit_hessian");
    return 0;
}

extern "C" int oovqe_circuit_hessian(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                     int n_gates, int n_qubits, int ncas, uint32_t init_index,
                                     const double* c1, const double* c2, const int32_t* pairs,
                                     int n_pairs, double* work, double* H, oovqe_stream_t stream)
{
    return oovqe_circuit_hessian_batched_impl(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index, c1, c2,
                                              0, 0, pairs, n_pairs, 1, work, H, n_theta, 0, stream);
}

extern "C" int oovqe_circuit_hessian_batch(const double* theta, int n_theta, const oovqe_gate_t* gates,
                                           int n_gates, int n_qubits, int ncas, uint32_t init_index,
                                           const double* c1, const double* c2, const int32_t* pairs,
                                           int n_pairs, int batch, double* work, double* H,
                                           oovqe_stream_t stream)
{
    const long na2 = (long)ncas * ncas;
    return oovqe_circuit_hessian_batched_impl(theta, n_theta, gates, n_gates, n_qubits, ncas, init_index, c1, c2,
                                              na2, na2 * na2, pairs, n_pairs, batch, work, H, n_theta,
                                              (long)n_theta * n_theta, stream);
}
