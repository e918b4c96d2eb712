// Spin-orbital (unrestricted) RDMs of a statevector.
//
// Parameterized_circuit.get_rdms_from_state(state, restricted=False) of the reference
// (src/auto_oo/pqc.py:192-218 with the unrestricted operators of utils/active_space.py:29-83:
// e_pq = a+_p a_q, e_pqrs = a+_p a+_q a_r a_s over the 2 ncas spin orbitals, Jordan-Wigner with
// mode j <-> wire j, wire 0 the most significant bit):
//   gamma[p,q]     = sum_x bra[y] s ket[x],   |y> s = a+_p a_q |x>
//   Gamma[p,q,r,s] = sum_x bra[y] s ket[x],   |y> s = a+_p a+_q a_r a_s |x>
// (bilinear form, no conjugation -- the caller combines real and imaginary parts).
// Off the hot path: one workgroup per matrix element, the ladder operators applied by bit
// operations on the basis index, a fixed-shape tree reduction over the basis states.
#include "common.h"

namespace {

constexpr int SPIN_THREADS = 256;

// a_j (create == false) or a+_j (create == true) on basis state x of an n-qubit register;
// returns false if the result vanishes, else updates x and multiplies sign by the JW phase
__device__ __forceinline__ bool ladder(unsigned& x, int j, int n, bool create, int& sign)
{
    const unsigned bit = 1u << (n - 1 - j);
    const bool occ = (x & bit) != 0;
    if (occ == create) return false;
    const unsigned below = ~((bit << 1) - 1u);              // modes k < j sit on the higher bits
    if (__popc(x & below) & 1) sign = -sign;
    x ^= bit;
    return true;
}

__global__ __launch_bounds__(SPIN_THREADS)
void spin_rdm_kernel(const double* __restrict__ bra, const double* __restrict__ ket, int n,
                     double* __restrict__ gamma, double* __restrict__ Gamma)
{
    __shared__ double red[SPIN_THREADS];
    const unsigned D = 1u << n;
    const int n2 = n * n;
    const long e = blockIdx.x;                      // element: first the n^2 of gamma, then the n^4 of Gamma
    bra += (size_t)blockIdx.y * D;
    ket += (size_t)blockIdx.y * D;
    const bool one = e < n2;
    int p, q, r = 0, s = 0;
    if (one) { p = (int)(e / n); q = (int)(e - (long)p * n); }
    else {
        long f = e - n2;
        s = (int)(f % n); f /= n;
        r = (int)(f % n); f /= n;
        q = (int)(f % n);
        p = (int)(f / n);
    }
    double acc = 0.0;
    for (unsigned x0 = threadIdx.x; x0 < D; x0 += SPIN_THREADS) {
        unsigned x = x0;
        int sign = 1;
        bool ok;
        if (one) ok = ladder(x, q, n, false, sign) && ladder(x, p, n, true, sign);
        else ok = ladder(x, s, n, false, sign) && ladder(x, r, n, false, sign) &&
                  ladder(x, q, n, true, sign) && ladder(x, p, n, true, sign);
        if (ok) acc += bra[x] * (double)sign * ket[x0];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = SPIN_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (one) gamma[(size_t)blockIdx.y * n2 + e] = red[0];
        else Gamma[(size_t)blockIdx.y * n2 * n2 + (e - n2)] = red[0];
    }
}

}  // namespace

extern "C" int oovqe_spin_rdms(const double* bra, const double* ket, int n_qubits, int batch,
                               double* gamma, double* Gamma, oovqe_stream_t stream)
{
    OOVQE_REQUIRE(bra && ket && gamma && Gamma, "oovqe_spin_rdms: null pointer");
    OOVQE_REQUIRE(n_qubits >= 1 && n_qubits <= 20 && batch >= 1 && batch <= 65535,
                  "oovqe_spin_rdms: n_qubits=%d batch=%d", n_qubits, batch);
    const long n2 = (long)n_qubits * n_qubits;
    hipLaunchKernelGGL(spin_rdm_kernel, dim3((unsigned)(n2 + n2 * n2), (unsigned)batch), dim3(SPIN_THREADS), 0,
                       (hipStream_t)stream, bra, ket, n_qubits, gamma, Gamma);
    OOVQE_CHECK_LAUNCH("oovqe_spin_rdms");
    return 0;
}
