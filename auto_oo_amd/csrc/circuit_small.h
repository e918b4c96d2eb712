// One-workgroup circuit + RDM evaluation for small active spaces, shared by circuit.hip (its own
// launch) and contract.hip (extra workgroups of a K1 launch: the circuit of an evaluation is
// independent of the integral transform until the Fock stage, and a separate launch costs ~12 us
// of pure latency).
#pragma once
#include "common.h"

static __device__ __forceinline__ uint32_t deposit(uint32_t t, const oovqe_gate_t& g)
{
    // insert zero bits at the (ascending) positions g.pos[0..nfix)
    // (fixed trip count with constant indices: a runtime index into g.pos would put the gate in
    // scratch memory)
    uint32_t x = t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < g.nfix) {
            const uint32_t p = (uint32_t)g.pos[i];
            const uint32_t low = x & ((1u << p) - 1u);
            x = ((x >> p) << (p + 1)) | low;
        }
    }
    return x;
}


// ------------------------------------------------------------------------------------------
// Small active spaces (everything fits one workgroup's LDS, e.g. CAS(4e,3o): D = 64, 5 vectors):
// state + tangents + E_pq applications + all RDM sets in ONE launch.
//   phase 1  each wave runs the circuit for its vectors (psi, d psi/d theta_k) in LDS
//   phase 2  V[v][pq][x] = (E_pq vec_v)[x] by bit operations
//   phase 3  Gram products on the f64 MFMA: for set k
//              G_k[m][n] = sum_x A_k[m][x] B_0[n][x] + A_0[m][x] B_k[n][x]
//            rows m < a^2: A_v[m=pq] = V_v[qp];  row m = a^2: A_v = vec_v;  cols: B_v[n=rs] = V_v[rs]
//            gamma_k[rs] = G_k[a^2][rs];  Gamma_k[pq,rs] = G_k[pq][rs] - delta_qr gamma_k[ps]
// ------------------------------------------------------------------------------------------
constexpr int SMALL_THREADS = 512;   // = the K1 workgroup size (contract.hip)

static __device__ void apply_gate_wave(double* st, uint32_t D, const oovqe_gate_t& g, double c, double s,
                                bool deriv, int lane)
{
    const uint32_t fm = g.mask_hi | g.mask_lo;
    const uint32_t npairs = D >> g.nfix;
    const double h = 0.5 * (double)g.sign;
    for (uint32_t t = lane; t < npairs; t += 64) {
        const uint32_t x = deposit(t, g) | g.mask_hi;
        const uint32_t y = x ^ fm;
        const double pi = (__popc(x & g.mask_par) & 1) ? -1.0 : 1.0;
        const double ax = st[x], ay = st[y];
        if (!deriv) {
            st[x] = c * ax + pi * s * ay;
            st[y] = c * ay - pi * s * ax;
        } else {
            st[x] = h * (-s * ax + pi * c * ay);
            st[y] = h * (-s * ay - pi * c * ax);
        }
    }
    if (deriv) {
        for (uint32_t x = lane; x < D; x += 64) {
            const uint32_t f = x & fm;
            if (f != g.mask_hi && f != g.mask_lo) st[x] = 0.0;
        }
    }
}

// b = index of the circuit instance (geometry of a batch); lds = the workgroup's dynamic LDS
// (oovqe_small_circuit_lds_bytes() bytes); must be entered by all SMALL_THREADS threads.
static __device__ void circuit_rdm_small_body(const double* __restrict__ theta, int n_theta,
                                              const oovqe_gate_t* __restrict__ gates, int n_gates,
                                              int n_qubits, int ncas, uint32_t init_index, int n_tan,
                                              double* __restrict__ psi_out, double* __restrict__ dpsi_out,
                                              double* __restrict__ gamma, double* __restrict__ Gamma,
                                              const int b, double* lds)
{
    const uint32_t D = 1u << n_qubits;
    const int LDV = D + 2;                      // pitch = 2 mod 32 doubles: conflict-free ds_read_b64
    const int na2 = ncas * ncas;
    const int nvec = 1 + n_tan;
    const int nrow = na2 + 1;                   // A rows: a^2 operators + the vector itself
    const int MT = (nrow + 15) / 16, NT = (na2 + 15) / 16;
    double* vec = lds;                                   // [nvec][LDV]
    double* V = vec + (size_t)nvec * LDV;                // [nvec][na2][LDV]
    double* R = V + (size_t)nvec * na2 * LDV;            // [2*nvec][MT*16][NT*16]
    const int RSZ = MT * 16 * NT * 16;
    double* cs_l = R + (size_t)2 * nvec * RSZ;           // [n_gates][2] cos, sin
    oovqe_gate_t* gl = reinterpret_cast<oovqe_gate_t*>(cs_l + 2 * n_gates);   // [n_gates]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const double* th = theta + (size_t)b * n_theta;
    // gate table, cos/sin of every gate angle: once per workgroup, then LDS only
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gates);
        uint32_t* dst = reinterpret_cast<uint32_t*>(gl);
        for (int idx = tid; idx < n_gates * (int)(sizeof(oovqe_gate_t) / 4); idx += SMALL_THREADS)
            dst[idx] = src[idx];
        for (int g = tid; g < n_gates; g += SMALL_THREADS) {
            const int ti = gates[g].theta_idx;
            double sn = 0.0, cs = 1.0;
            if (ti >= 0) sincos(0.5 * (double)gates[g].sign * th[ti], &sn, &cs);
            cs_l[2 * g] = cs;
            cs_l[2 * g + 1] = sn;
        }
    }
    __syncthreads();

    // ---- phase 1: circuits (one vector per wave at a time; a tangent whose parameter drives
    // several gates is the sum over those gates, accumulated in place) -------------------------
    for (int v = wave; v < nvec; v += SMALL_THREADS / 64) {
        double* st = vec + (size_t)v * LDV;
        const int k = v - 1;
        bool first = true;
        int g_start = 0;
        while (true) {
            int dgate = -1;
            if (k >= 0) {
                for (int g = g_start; g < n_gates; ++g)
                    if (gl[g].theta_idx == k) { dgate = g; break; }
                if (dgate < 0) break;
                g_start = dgate + 1;
            }
            // run into registers-free scratch: reuse st when this is the first occurrence,
            // otherwise run in the V area of this vector (free until phase 2) and add
            double* w = first ? st : V + (size_t)v * na2 * LDV;
            for (uint32_t x = lane; x < D; x += 64) w[x] = (x == init_index) ? 1.0 : 0.0;
            __builtin_amdgcn_wave_barrier();
            for (int g = 0; g < n_gates; ++g) {
                const oovqe_gate_t gt = gl[g];
                if (gt.theta_idx < 0) continue;
                apply_gate_wave(w, D, gt, cs_l[2 * g], cs_l[2 * g + 1], g == dgate, lane);
                __builtin_amdgcn_wave_barrier();
            }
            if (!first)
                for (uint32_t x = lane; x < D; x += 64) st[x] += w[x];
            first = false;
            if (k < 0) break;
        }
        if (k >= 0 && first)
            for (uint32_t x = lane; x < D; x += 64) st[x] = 0.0;
    }
    __syncthreads();
    if (psi_out)
        for (uint32_t x = tid; x < D; x += SMALL_THREADS) psi_out[(size_t)b * D + x] = vec[x];
    if (dpsi_out)
        for (int idx = tid; idx < n_tan * (int)D; idx += SMALL_THREADS) {
            const int k = idx / D, x = idx - k * D;
            dpsi_out[((size_t)b * n_tan + k) * D + x] = vec[(size_t)(1 + k) * LDV + x];
        }

    // ---- phase 2: V[v][pq][x] ------------------------------------------------------------------
    for (int idx = tid; idx < nvec * na2 * (int)D; idx += SMALL_THREADS) {
        const uint32_t x = idx % D;
        const int vp = idx / D;
        const int pq = vp % na2, v = vp / na2;
        const int p = pq / ncas, q = pq - p * ncas;
        const double* src = vec + (size_t)v * LDV;
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
        V[((size_t)v * na2 + pq) * LDV + x] = acc;
    }
    __syncthreads();

    // ---- phase 3: Gram products.  unit u = 2k + h: h = 0 -> A_k . B_0, h = 1 -> A_0 . B_k ------
    const int RS = RSZ;
    const int ntile = MT * NT;
    const int nunits = 2 * nvec * ntile;
    for (int uu = wave; uu < nunits; uu += SMALL_THREADS / 64) {
        const int u = uu / ntile, tile = uu - u * ntile;
        const int k = u >> 1, h = u & 1;
        const int mt = tile / NT, nt = tile - mt * NT;
        double* Ru = R + (size_t)u * RS;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (!(k == 0 && h == 1)) {
            const int va = h == 0 ? k : 0, vb = h == 0 ? 0 : k;
            // A row for this lane: m = mt*16 + lr
            const int m = mt * 16 + lr;
            const double* arow = nullptr;
            if (m < na2) {
                const int p = m / ncas, q = m - p * ncas;
                arow = V + ((size_t)va * na2 + q * ncas + p) * LDV;      // V_va[qp]
            } else if (m == na2) {
                arow = vec + (size_t)va * LDV;
            }
            const int n = nt * 16 + lr;
            const double* brow = n < na2 ? V + ((size_t)vb * na2 + n) * LDV : nullptr;
            // 16 k-steps at a time: all fragment reads first, then the MFMA chain (one LDS latency
            // per 16 MFMAs instead of one per MFMA); D is a power of two >= 4
            for (uint32_t x0 = 0; x0 < D; x0 += 64) {
                double af[16], bf[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t x = x0 + 4 * i + lq;
                    const bool in = x0 + 4 * i < D;
                    af[i] = (arow && in) ? arow[x] : 0.0;
                    bf[i] = (brow && in) ? brow[x] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = mfma_f64(af[i], bf[i], acc);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            Ru[(mt * 16 + lq + 4 * i) * (NT * 16) + nt * 16 + lr] = acc[i];
    }
    __syncthreads();
    const int ldr = NT * 16;
    for (int idx = tid; idx < nvec * na2; idx += SMALL_THREADS) {
        const int k = idx / na2, rs = idx - k * na2;
        const double g = R[(size_t)(2 * k) * RS + na2 * ldr + rs] +
                         R[(size_t)(2 * k + 1) * RS + na2 * ldr + rs];
        gamma[((size_t)b * nvec + k) * na2 + rs] = g;
    }
    for (int idx = tid; idx < nvec * na2 * na2; idx += SMALL_THREADS) {
        const int k = idx / (na2 * na2);
        const int rem = idx - k * na2 * na2;
        const int pq = rem / na2, rs = rem - pq * na2;
        const int p = pq / ncas, q = pq - p * ncas;
        const int r = rs / ncas, s2 = rs - r * ncas;
        double g = R[(size_t)(2 * k) * RS + pq * ldr + rs] + R[(size_t)(2 * k + 1) * RS + pq * ldr + rs];
        if (q == r) {
            const int ps = p * ncas + s2;
            g -= R[(size_t)(2 * k) * RS + na2 * ldr + ps] + R[(size_t)(2 * k + 1) * RS + na2 * ldr + ps];
        }
        Gamma[((size_t)b * nvec + k) * na2 * na2 + rem] = g;
    }
}


// LDS bytes of circuit_rdm_small_body
static inline size_t oovqe_small_circuit_lds_bytes(int n_qubits, int ncas, int nvec, int n_gates)
{
    const size_t D = (size_t)1 << n_qubits;
    const size_t LDV = D + 2;
    const int na2 = ncas * ncas;
    const int MT = (na2 + 1 + 15) / 16, NT = (na2 + 15) / 16;
    return ((size_t)nvec * LDV + (size_t)nvec * na2 * LDV + (size_t)2 * nvec * MT * 16 * NT * 16 +
            (size_t)2 * n_gates) * sizeof(double) + (size_t)n_gates * sizeof(oovqe_gate_t);
}
